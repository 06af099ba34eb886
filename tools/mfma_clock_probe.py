#!/usr/bin/env python
"""Sustained issue rate of v_mfma_f32_32x32x16_f16 with every CU busy (tools only): the ceiling the conv / GEMM loops can reach."""
import ctypes, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "stamp", "libstamp.so"))
lib.mfma_rate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(2, dtype=torch.int64, device=dev); sink = torch.zeros(1, device=dev)
for waves_per_simd, blocks in ((1, 256), (2, 256), (3, 256), (4, 256), (1, 32), (2, 32), (4, 32)):
    iters = 20000
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.mfma_rate(out.data_ptr(), iters, blocks, 256 * waves_per_simd, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
    ticks, cyc = out.tolist()
    sec = e0.elapsed_time(e1) * 1e-3            # whole kernel (the oldest wave of a SIMD is served first: its own loop time says nothing)
    cyc = cyc * (sec / (ticks / 1e8))           # shader cycles scaled to the kernel duration (clock estimate only)
    n = iters * 8 * waves_per_simd                      # MFMAs per SIMD
    print(f"{blocks} blocks x {waves_per_simd} wave/SIMD: {sec*1e6:8.1f} us, {n/sec/1e6:7.1f} M MFMA/s/SIMD = {sec/n*2.4e9:5.1f} cycles@2.4GHz per MFMA, "
          f"s_memtime clock {cyc/sec/1e9:.2f} GHz, chip-wide {blocks/256*256*4*n/sec*32768/1e15:.2f} PFLOP/s")
