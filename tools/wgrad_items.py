"""Where the time of the grouped weight-gradient launch goes, per kind of work item: every workgroup of k_wgrad_group logs its kind and its
first / last 100 MHz tick (a -DWGG_ITEMLOG build of csrc/wgrad.hip: bash tools/build_variant.sh wgg_items wgrad.hip -DWGG_ITEMLOG), this script
runs one eager training step on it and prints, per kind, the items, their mean / longest duration, and when the kind's first item started and
its last one ended (us from the launch's first tick).
usage: NDJIR_HIP_LIB=ndjir_amd/_lib/variants/wgg_items.so python tools/wgrad_items.py [config]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ndjir_amd import config as cfg
from ndjir_amd import lib
from ndjir_amd.step import Step

name = sys.argv[1] if len(sys.argv) > 1 else "default"
lib.load()
step = Step(cfg.load(name, []), 512, torch.device("cuda", 0), 0, 1)
for _ in range(3):
    step.forward_backward()
torch.cuda.synchronize()
h = lib.load()
buf = np.zeros((16384, 3), dtype=np.int64)
rc = h.ndjir_debug_wgrad_items(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
for lo, hi, title in ((8192, 16384, "k_wgrad_group_wide"), (0, 8192, "k_wgrad_group")):
  part = np.zeros_like(buf[:, 0], dtype=bool); part[lo:hi] = True
  live = (buf[:, 2] > 0) & part
  if not live.any():
      continue
  print("##", title)
  t0 = buf[live, 1].min()
  names = {5: "128x256 item (wide launch)", 0: "128x128 tile", 1: "32x128 strip", 2: "128x32 strip", 3: "narrow output", 4: "64x128 item"}
  print("kind                               items   mean us   max us   first start   last end   item-us")
  for k in sorted(set(buf[live, 0])):
      m = live & (buf[:, 0] == k)
      d = (buf[m, 2] - buf[m, 1]) / 100.0
      nm = names.get(int(k) & 15, str(int(k) & 15)) + " lay %d%s" % ((int(k) >> 4) & 3, " P=131072" if int(k) & 64 else "")
      print("%-34s %5d  %8.1f %8.1f   %10.1f %10.1f  %9.0f" % (nm, m.sum(), d.mean(), d.max(), (buf[m, 1].min() - t0) / 100.0,
                                                               (buf[m, 2].max() - t0) / 100.0, d.sum()))
  print("launch: %.1f us, %d workgroups" % ((buf[live, 2].max() - t0) / 100.0, live.sum()))
