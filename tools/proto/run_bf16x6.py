import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libproto.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       os.path.join(here, "bf16x6_gemm.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K, scale) in ((256, 512, 1152, 1.0), (256, 512, 2304, 1.0), (128, 256, 576, 100.0)):
    A = (torch.randn(M, K, device=dev) * scale).contiguous(); B = torch.randn(N, K, device=dev).contiguous()
    ref = (A.double() @ B.double().t())
    out = {}
    for mode in (0, 1):
        C = torch.empty(M, N, device=dev)
        lib.proto_gemm(mode, ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B.data_ptr()), ctypes.c_void_p(C.data_ptr()), M, N, K, None)
        torch.cuda.synchronize()
        err = (C.double() - ref).abs()
        out[mode] = (err.max().item() / ref.abs().max().item(), (err / (A.double().abs() @ B.double().abs().t())).max().item())
    print("M%d N%d K%d: bf16x6 maxerr/maxref %.3e  err/sum|ab| %.3e | f32mfma %.3e %.3e" % (M, N, K, out[0][0], out[0][1], out[1][0], out[1][1]))
blocks, iters = 256 * 8, 4000
buf = torch.empty(blocks * 256, device=dev)
for mode, flops in ((0, 2 * 32 * 32 * 16), (1, 2 * 32 * 32 * 2)):
    lib.proto_rate(mode, ctypes.c_void_p(buf.data_ptr()), blocks, 10, None); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); lib.proto_rate(mode, ctypes.c_void_p(buf.data_ptr()), blocks, iters, None); e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    tf = blocks * 4 * iters * 4 * flops / ms / 1e9
    print("mode %d raw MFMA rate: %.1f TF/s (%.3f ms)%s" % (mode, tf, ms, "  -> /6 = %.1f f32-equivalent TF" % (tf / 6) if mode == 0 else ""))
