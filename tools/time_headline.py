"""The three headline launches of bench.py on their own (fp64 table, closed form, 16-bit table), kernel times from
HIP events: time_headline.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-qubo", "--steps", "30", "--warmup", "5"],
                         capture_output=True, text=True, check=True, cwd=ROOT).stdout
    d = json.loads(out.strip().splitlines()[-1])
    for k, v in d["roofline_by_weights"].items():
        print(k, v["kernel"], round(v["kernel_ms"], 4), round(v["frac"], 4), flush=True)


if __name__ == "__main__":
    main()
