// Shared by the translation units of libqplace.so; not part of the public ABI.
#pragma once
#include "../../include/qplace.h"

namespace qp {
// Records a printf-style message for qp_last_error() and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace qp
