"""Phase timeline of the fused MLP chain kernel (workgroup 0, first tile): shader-clock stamps per
layer and wave -> where the cycles of a layer go (k-loop / staging / epilogue / barrier wait).
usage: python tools/chain_timeline.py [fwd|bwd]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndjir_amd import lib  # noqa: E402
from ndjir_amd.mlp import chain_forward, fused_mlp  # noqa: E402
from kernel_bench import make  # noqa: E402


def show(buf, L, title):
    t = buf.cpu().numpy().reshape(10, 5, 8)[:L].astype(np.int64)
    live = t[0, 0] > 0                    # (4-wave kernels leave the slots of waves 4..7 empty)
    t = t[:, :, live]
    t0 = t[0, 0].min()
    print(title)
    print("layer |  start  | k-loop (min/max over waves) | stage | epilogue | barrier wait (min/max) | layer total")
    for li in range(L):
        s, k, p1, p2, b = (t[li, i] for i in range(5))
        if k.max() == 0:      # narrow layer: no unit stamps
            print(f"  {li}   | {s.min() - t0:7d} | (narrow path)  total {b.max() - s.min():6d}")
            continue
        act = k > 0
        print(f"  {li}   | {s.min() - t0:7d} | {(k - s)[act].min():6d} {(k - s)[act].max():6d} | "
              f"{(p1 - k)[act].mean():6.0f} | {(p2 - p1)[act].mean():6.0f} | {(b - p2).min():6d} {(b - p2).max():6d} | "
              f"{b.max() - s.min():6d}")
    print(f"total {t[L - 1, 4].max() - t0} cycles")
    full = buf.cpu().numpy().reshape(10, 5, 8).astype(np.int64)[9]
    if full[0].max() > 0:
        k0 = full[0].min()
        rt = buf.cpu().numpy().reshape(10, 5, 8).astype(np.int64)[8]
        if rt[0].max() > 0:
            ticks = rt[1].max() - rt[0].min()
            print(f"workgroup 0 lifetime: {full[3].max() - k0} shader cycles in {ticks} ticks of the 100 MHz clock "
                  f"-> {100e6 * (full[3].max() - k0) / max(ticks, 1) / 1e9:.2f} GHz effective shader clock")
        print(f"workgroup 0: kernel start -> input stage done {full[1].max() - k0}, -> first tile done {full[2].max() - k0}, "
              f"-> all its tiles done {full[3].max() - k0} cycles")
    if os.environ.get("TIMELINE_RAW"):
        li = int(os.environ["TIMELINE_RAW"])
        print(f"layer {li} per wave (relative to layer start): k-loop done, staged, epilogue done, barrier passed")
        for w in range(8):
            print(f"  wave {w}: " + " ".join(f"{t[li, i, w] - t[li, 0].min():7d}" for i in range(5)))


BLOCKED = bool(os.environ.get("TIMELINE_BLOCKED"))      # point-blocked side tensors: the pipelined kernel (mlp3p.hip) takes the launch


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    P = int(os.environ.get("TIMELINE_P", "65536"))      # (fewer points than 128 x CUs: NDJIR_MLP_TILE=128 keeps the 128-point-tile kernel)
    dims = (43, 256, 256, 256, 213, 256, 256, 256, 257)
    Ws, bs = make(dims, 1, 3)
    x = torch.randn(P, 43, device="cuda")
    NREC = int(os.environ.get("TIMELINE_BLOCKS", "0"))
    buf = torch.zeros(10 * 5 * 8 + 3 * NREC, dtype=torch.int64, device="cuda")
    buf[399] = NREC
    so = lib.load()
    so.ndjir_mlp_debug_timeline.argtypes = [ctypes.c_void_p]
    if mode in ("fwd", "fwd_nostore"):
        keep = mode == "fwd"
        chain_forward(x, Ws, bs, 100.0, 3, 0.7071, keep_hidden=keep, blocked=BLOCKED)
        torch.cuda.synchronize()
        so.ndjir_mlp_debug_timeline(buf.data_ptr())
        reps = int(os.environ.get("TIMELINE_REPS", "1"))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):      # > 1: stamps of the last of a back-to-back series
            chain_forward(x, Ws, bs, 100.0, 3, 0.7071, keep_hidden=keep, blocked=BLOCKED)
        e1.record()
        torch.cuda.synchronize()
        print(f"{reps} launches: {e0.elapsed_time(e1) * 1e3 / reps:.1f} us per launch by HIP events")
        so.ndjir_mlp_debug_timeline(None)
        show(buf[:400], 8, "geometric net forward chain (43-256-256-256-213|+43-256-256-256-257), tile of 64 points")
        if NREC >= 64 and BLOCKED and os.environ.get("TIMELINE_SUB"):
            # diagnostic build of mlp3p.hip (-DNDJIR_CHAINP_SUBSTAMP): per-slot stamps of layer 2
            sub = buf[400:400 + 72].cpu().numpy().reshape(4, 18).astype(np.int64)
            for ph in range(2):
                for wv in range(2):
                    r = sub[ph * 2 + wv]
                    if r[0] > 0:
                        print(f"phase (2, {ph}) wave {4 * wv}: cycles per slot", " ".join(f"{int(b - a_):5d}" for a_, b in zip(r[:-1], r[1:-1] if False else r[1:17])))
            return
        if NREC:
            rec = buf[400:].cpu().numpy().reshape(NREC, 3)
            t0 = rec[:, 0].min()
            st, en, hw = (rec[:, 0] - t0) / 100.0, (rec[:, 1] - t0) / 100.0, rec[:, 2]
            print(f"{NREC} workgroups: start min/median/max {st.min():.1f}/{np.median(st):.1f}/{st.max():.1f} us, "
                  f"duration min/median/max {(en - st).min():.1f}/{np.median(en - st):.1f}/{(en - st).max():.1f} us, last end {en.max():.1f} us")
            order = np.argsort(st)
            for q in (0, 255, 256, 511, 512, 767, 768, 1023):
                if q < NREC:
                    b = order[q]
                    print(f"  {q}-th to start: block {b} start {st[b]:.1f} dur {en[b] - st[b]:.1f} hw_id 0x{int(hw[b]):x}")
            ids = {}
            for b in range(NREC):
                ids.setdefault(int(hw[b]) & 0xfffffff0, []).append(b)     # drop the wave-slot bits
            cnt = np.array([len(v) for v in ids.values()])
            print(f"  distinct (se, sh, cu, simd...) ids {len(ids)}; workgroups per id min/max {cnt.min()}/{cnt.max()}")
    elif mode == "sub":          # sub-phase stamps of layer 2 (tools/build_variant.sh sub mlp3w.hip -DNDJIR_CHAIN_SUBSTAMP, NDJIR_HIP_LIB=...): slot 5 = before / after each row block
        dims2 = (259, 256, 256, 256, 3)
        W2, b2 = make(dims2, 2)
        xg = torch.randn(P, dims2[0], device="cuda")
        which = os.environ.get("SUB_MODE", "fwd")
        Wg = [w.clone().requires_grad_(which == "bwd") for w in W2]
        bg = [b.clone().requires_grad_(which == "bwd") for b in b2]
        if which == "fwd":
            chain_forward(xg, W2, b2, keep_hidden=True)
            torch.cuda.synchronize()
            so.ndjir_mlp_debug_timeline(buf.data_ptr())
            chain_forward(xg, W2, b2, keep_hidden=True)
        else:
            y = fused_mlp(xg, Wg, bg)
            gy = torch.randn_like(y)
            torch.cuda.synchronize()
            so.ndjir_mlp_debug_timeline(buf.data_ptr())
            torch.autograd.grad(y, Wg + bg, gy)
        torch.cuda.synchronize()
        so.ndjir_mlp_debug_timeline(None)
        t = buf[:400].cpu().numpy().reshape(10, 5, 8).astype(np.int64)
        print(which, "layer 2: k-loop end -> [before block 0, after 0, after 1, after 2, after 3] -> phase A end; per wave, relative to the layer's start")
        for w in range(8):
            print(f"  wave {w}: kloop {t[2, 1, w] - t[2, 0].min():6d} | " + " ".join(f"{t[5, i, w] - t[2, 0].min():6d}" for i in range(5)) + f" | A end {t[2, 2, w] - t[2, 0].min():6d}  barrier {t[2, 3, w] - t[2, 0].min():6d}  end {t[2, 4, w] - t[2, 0].min():6d}")
    else:
        dims2 = (259, 256, 256, 256, 3) if mode == "bwd" else (39, 128, 128, 128, 1)
        W2, b2 = make(dims2, 2)
        xg = torch.randn(P, dims2[0], device="cuda").requires_grad_(mode == "bwd")
        Wg = [w.clone().requires_grad_(True) for w in W2]
        bg = [b.clone().requires_grad_(True) for b in b2]
        y = fused_mlp(xg, Wg, bg)
        gy = torch.randn_like(y)
        torch.cuda.synchronize()
        so.ndjir_mlp_debug_timeline(buf.data_ptr())
        torch.autograd.grad(y, ([xg] if mode == "bwd" else []) + Wg + bg, gy)
        torch.cuda.synchronize()
        so.ndjir_mlp_debug_timeline(None)
        show(buf, 4 if mode == "bwd" else 3, f"backward chain of net {dims2}")


if __name__ == "__main__":
    main()
