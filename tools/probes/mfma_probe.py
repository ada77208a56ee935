"""Sustained bf16 MFMA rate without memory traffic (see mfma_probe.hip); run on the GPU box."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "mfma_probe.so"))
dev = torch.device("cuda:0")
sink = torch.zeros(1 << 16, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for shape, per_iter, rnd in ((16, 16 * 16 * 16 * 32 * 2, 0), (32, 8 * 32 * 32 * 16 * 2, 0), (16, 16 * 16 * 16 * 32 * 2, 1), (32, 8 * 32 * 32 * 16 * 2, 1)):
    for wpc in (4, 8):  # waves per CU
        blocks = 256 * wpc // 4
        iters = (400000 if wpc == 4 else 200000) // (2 if shape == 32 else 1)
        run = lambda: lib.mfma_run(ctypes.c_void_p(sink.data_ptr()), blocks, iters, shape, rnd, ctypes.c_void_p(st))
        run(); torch.cuda.synchronize()
        rates = []
        for rep in range(12):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(); e.record(); torch.cuda.synchronize()
            t = s.elapsed_time(e) * 1e-3
            rates.append(blocks * 4 * iters * per_iter / t / 1e12)
        print(f"mfma {shape}x{shape} {'random' if rnd else 'constant'} operands, {wpc} waves/CU: launch {t * 1e3:.0f} ms; TFLOP/s per launch: " + " ".join(f"{r:.0f}" for r in rates), flush=True)
