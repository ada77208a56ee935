"""Does the ORDER in which a 16x16x32 MFMA stream presents its operands change what the power-capped chip sustains?  (see mfma_probe.hip)"""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "mfma_probe.so"))
sink = torch.zeros(1 << 16, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
per_iter, blocks, iters = 16 * 16 * 16 * 32 * 2, 512, 200000
names = {16: "both operands change every MFMA", 161: "first operand held for 4 MFMAs", 162: "second operand held for 4 MFMAs", 163: "GEMM quad order (first alternates, second held for 2)"}
for rep in range(2):
    for shape in (16, 161, 162, 163):
        rates = []
        for i in range(6):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); lib.mfma_run(ctypes.c_void_p(sink.data_ptr()), blocks, iters, shape, 1, ctypes.c_void_p(st)); e.record(); torch.cuda.synchronize()
            if i:
                rates.append(blocks * 4 * iters * per_iter / (s.elapsed_time(e) * 1e-3) / 1e12)
        print(f"16x16x32, 8 waves/CU, random operands, {names[shape]:55s}: " + " ".join(f"{r:.0f}" for r in rates), flush=True)
