"""Driver of tools/probes/pk_victim.hip: the packed-vs-scalar victim kernel on stream A, alone and next to a library
kernel looping on stream B (aggressors taken from a traced forward: a 3x3 halo convolution, a row-GEMM, the level-0 scan)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from founddiff_amd import _lib as L, synth
from founddiff_amd.engine import DAEngine
dev = torch.device("cuda")
V = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libpkvictim.so"))
V.pk_victim_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
V.pk_aggressor_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
B = 8
dif, w = bench.build_model(dev)
sd = {k[len("model.unet0."):]: v for k, v in w.items() if k.startswith("model.unet0.")}
eng = DAEngine(sd, "", dev, "bf16")
_, ld = synth.ct_phantom(B, 512, seed=10)
x_in = (torch.from_numpy(ld).to(dev) * 2 - 1).contiguous()
img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
tb = torch.full((B,), 500.0, device=dev)
eng.encode_condition(x_in)
eng.forward(img, x_in, tb)
L.TRACE = []
eng.forward(img, x_in, tb)
trace, L.TRACE = L.TRACE, None
lib = L.lib()
torch.cuda.synchronize()


def pick(pred):
    return [(n, a) for n, a in trace if pred(n, a)][0]


aggr = {
    "none": None,
    "own: MFMA on registers": ("own", 1),
    "own: LDS traffic": ("own", 2),
    "own: MFMA + LDS": ("own", 3),
    "conv3x3 halo 512->512 @64 (library)": pick(lambda n, a: n == "fd_conv2d" and lib.fd_conv_kernel_id(a[0]) == 11 and a[0]._obj.c0 == 512 and a[0]._obj.Cout == 512),
    "conv3x3 weights-in-registers 64->64 @512": pick(lambda n, a: n == "fd_conv2d" and lib.fd_conv_kernel_id(a[0]) == 13),
    "row-GEMM (z recompute) @512": pick(lambda n, a: n == "fd_conv2d" and a[0]._obj.prologue == 3),
    "pw_gemm 512->2048 @64": pick(lambda n, a: n == "fd_conv2d" and lib.fd_conv_kernel_id(a[0]) == 7),
    "scan level 0": pick(lambda n, a: n == "fd_selective_scan_xproj"),
    "fd_pw_dw3x3 @512": pick(lambda n, a: n == "fd_pw_dw3x3"),
    "fd_dwconv3x3": pick(lambda n, a: n == "fd_dwconv3x3"),
}
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
blocks, iters = 256 * 4, 20000            # 4 workgroups of 4 waves per CU: one wave per SIMD per workgroup, room for the aggressor
out = torch.zeros(blocks * 256 * 2, dtype=torch.int32, device=dev)
names = {0: "v_pk_fma_f32 chain", 1: "v_pk_mul_f32 -> plain adds", 2: "v_exp x2 -> v_pk_mul / v_pk_fma", 3: "v_pk_add_f32 chain",
         4: "LDS row -> packed arithmetic", 5: "LDS row -> plain arithmetic", 6: "LDS row -> v_mov copies -> packed", 7: "LDS row, wait + s_nop 7 -> packed", 8: "GLOBAL row (global_load_dwordx4) -> packed"}
dummy = torch.zeros(64, device=dev)
grow = (0.9 + 0.001 * torch.arange(128, dtype=torch.float32)).repeat(2).to(dev)      # [0:128] feeds the LDS row and the reference, [128:256] mode 8
for key, la in aggr.items():
    for mode in range(9):
        out.zero_()
        torch.cuda.synchronize()
        if la is not None and la[0] == "own":
            with torch.cuda.stream(sB):
                V.pk_aggressor_launch(la[1], dummy.data_ptr(), 400000, 256 * 8, sB.cuda_stream)
        elif la is not None:
            n, a = la
            a = list(a)
            a[-1] = sB.cuda_stream
            with torch.cuda.stream(sB):
                for _ in range(40):
                    getattr(lib, n)(*a)
        with torch.cuda.stream(sA):
            V.pk_victim_launch(mode, out.data_ptr(), iters, blocks, grow.data_ptr(), sA.cuda_stream)
        torch.cuda.synchronize()
        o = out.view(-1, 2).cpu()
        lo, hi = int(o[:, 0].sum()), int(o[:, 1].sum())
        lanes = sorted(set(((o[:, 0] + o[:, 1]) > 0).nonzero().flatten().remainder(64).tolist()))
        print(f"aggressor {key:42s} victim {names[mode]:34s}: mismatches lo {lo:8d} hi {hi:8d}  lanes {lanes[:4]}..{lanes[-4:] if lanes else ''} ({len(lanes)})", flush=True)
