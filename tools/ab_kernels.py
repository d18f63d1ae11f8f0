"""Per-kernel-class averages of one bench run (for same-box A/B of the chain kernels): python tools/ab_kernels.py [bench args]"""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--train-steps", "0", "--steps", "20", "--warmup", "3"] + sys.argv[1:],
                     capture_output=True, text=True).stdout.strip().split("\n")[-1]
d = json.loads(out)
print(f"ms/step {d['ms_per_step']:.3f} (eager {d['eager']['ms_per_step']:.3f})  rays/s {d['value']:.0f}  loss {d['loss']:.10f}")
for k, v in d["kernels"].items():
    print(f"   {k:14s} {v['launches_per_step']:6.1f} x {v['avg_us']:8.1f} us = {v['ms_per_step']:6.3f} ms/step  {v['tflops']:6.1f} TFLOP/s")
print("   roofline:", d["roofline"]["kernel_class"], d["roofline"]["kernel"][:60], round(d["roofline"]["frac"], 3))
