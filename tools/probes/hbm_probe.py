"""Practical HBM read bandwidth (see hbm_probe.hip); run on the GPU box."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "hbm_probe.so"))
dev = torch.device("cuda:0")
buf = torch.empty(8 << 30, dtype=torch.uint8, device=dev).fill_(1)
sink = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
for nbytes in (64 << 20, 1 << 30, 8 << 30):
    for blocks in (1024, 2048, 4096, 8192):
        for unr in (2, 4, 8):
            st = torch.cuda.current_stream().cuda_stream
            run = lambda: lib.hbm_read(ctypes.c_void_p(buf.data_ptr()), ctypes.c_int64(nbytes), ctypes.c_void_p(sink.data_ptr()), blocks, unr, ctypes.c_void_p(st))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                run()
            e.record()
            torch.cuda.synchronize()
            t = s.elapsed_time(e) / 10
            print(f"{nbytes >> 20:5d} MB  blocks {blocks:5d}  {unr} loads in flight: {t * 1e3:8.1f} us  {nbytes / t / 1e9:6.2f} TB/s", flush=True)
