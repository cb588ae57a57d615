"""Where does a K-step of the convolution kernels go?  (diagnostic; GPU box)

    python tools/probes/igemm_phase_probe.py build [extra -D flags]     # here or on the box: lib_PROBE.so
    TV_HIP_SO=tools/probes/abl/lib_PROBE.so python tools/probes/igemm_phase_probe.py run [--halo 0|2|3] [--c 192] [--res 256] [--mb 64]

The probe build (-DTV_PROBE) brackets the phases of the main loop with s_memtime and sums, per wave, the shader cycles
spent   0 waiting for the DMA (vmcnt)   1 at the block barrier   2 issuing the next DMA   3 fragment reads (to lgkmcnt 0)
        4 MFMA issue   5 (loop exit)   6 epilogue.
Sixteen sampled blocks dump their counters; the timers serialise the phases a little, so read shares, not absolutes.
Ping-pong build (-DTV_HALO_PP=1): 1 = wait at the barrier that opens the load phase, 3 = fragment reads (issue + return),
2 = DMA issue, 0 = vmcnt wait, 5 ("exit") = wait at the barrier that opens the MFMA phase, 4 = the 48 MFMAs.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "deepl-project_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "probes", "abl")
NAMES = ["vmcnt wait", "barrier", "dma issue", "frag reads", "mfma", "exit", "epilogue", "-"]


def build(extra):
    """build [name=XYZ] [-D...]: lib_PROBE.so with the timers, or lib_XYZ.so with just the given defines (A/B timing)."""
    name = "PROBE"
    if extra and extra[0].startswith("name="):
        name, extra = extra[0][5:], extra[1:]
    if name == "PROBE":
        extra = ["-DTV_PROBE"] + extra
    os.makedirs(OUT, exist_ok=True)
    objs = []
    procs = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        o = f"/tmp/probe_{name}_{f[:-4]}.o"
        fl = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if f == "attention.hip" else []
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value"] + extra + fl + \
              ["-c", os.path.join(CSRC, f), "-o", o]
        procs.append(subprocess.Popen(cmd))
        objs.append(o)
    for p in procs:
        assert p.wait() == 0
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"lib_{name}.so")] + objs)
    print("built", os.path.join(OUT, f"lib_{name}.so"))


def run(argv):
    import argparse
    import ctypes as C
    ap = argparse.ArgumentParser()
    ap.add_argument("--halo", type=int, default=1)
    ap.add_argument("--c", type=int, default=192)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--mb", type=int, default=64)
    ap.add_argument("--cfg", default="")
    a = ap.parse_args(argv)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "deepl-project_amd"))
    import torch
    from transvae.hip import _lib, ops
    lib = _lib.load()
    lib.tv_set_igemm_halo(a.halo)
    if a.cfg:
        lib.tv_set_igemm_config(*[int(v) for v in a.cfg.split(",")])
    dev = torch.device("cuda:0")
    buf = torch.zeros(16 * 8 * 8, dtype=torch.int64, device=dev)
    lib.tv_set_igemm_probe.argtypes = [C.c_void_p]
    assert lib.tv_set_igemm_probe(C.c_void_p(buf.data_ptr())) == 0
    x = torch.randn(a.mb, a.res, a.res, a.c, device=dev).to(torch.bfloat16)
    w = torch.randn(a.c, 3, 3, a.c, device=dev) * (9 * a.c) ** -0.5
    for _ in range(3):
        ops.conv(x, w, None, None, mode="c3s1")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv(x, w, None, None, mode="c3s1")
    e1.record()
    torch.cuda.synchronize()
    t = buf.view(16, 8, 8).cpu().double()
    t = t[t.sum(dim=(1, 2)) > 0]
    tot = t.sum(-1).mean()
    print(f"halo={a.halo} c={a.c} res={a.res} mb={a.mb}: {e0.elapsed_time(e1) / 5:.3f} ms per launch (probe build), "
          f"{t.shape[0]} blocks sampled, {tot:.0f} cycles per block")
    print("  phase        mean cycles   share   | per wave (block 0)")
    for i, n in enumerate(NAMES[:7]):
        print(f"  {n:12s} {t[:, :, i].mean():11.0f}  {100 * t[:, :, i].mean() / tot:5.1f}%   | " + " ".join(f"{v:8.0f}" for v in t[0, :, i]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "run" else sys.argv[1:])
