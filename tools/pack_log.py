"""Which matrices are (re)packed during a steady-state step: python tools/pack_log.py [config]  (diagnostics)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd import config as cfg, lib  # noqa: E402
from ndjir_amd.step import Step  # noqa: E402


def main():
    conf = cfg.load(sys.argv[1] if len(sys.argv) > 1 else "default", [])
    dev = torch.device("cuda:0")
    step = Step(conf, 512, dev, 0, 1)
    for _ in range(2):
        step.compute()
    torch.cuda.synchronize()
    real = lib.call
    log = []

    def spy(name, *args):
        if name.startswith("mlp_pack"):
            log.append((name, [tuple(a.shape) if torch.is_tensor(a) else a for a in args[:1]], args[2:6] if name == "mlp_pack" else args[1:7]))
        return real(name, *args)
    lib.call = spy
    step.compute()
    torch.cuda.synchronize()
    lib.call = real
    for e in log:
        print(e)
    print(len(log), "pack launches in a steady-state step")


if __name__ == "__main__":
    main()
