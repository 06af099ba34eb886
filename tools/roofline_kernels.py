"""The three kernels bench.py prices against their rooflines, on the operands of the C2 workload (measurement harness only).

`build(model, vox, cfg)` returns a list of dicts {key, name, regex, bound, launch, flops, bytes}: `launch()` enqueues ONE launch of the
kernel on torch's current stream.  bench.py times them with hipEvents; tools/roofline_probe.py runs them under
`rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-include-regex <regex>` to produce profiles/r02_pmc.json.
"""
import torch

from bflow_amd import hip, split as S
from bflow_amd.corr import CorrBlockParallelMultiTarget, CorrComputation

CONV_NAME = "conv_halo_kernel<2,3,3> (encoder layer1 3x3 64->64, 5x240x320)"
K5_NAME = "corr_stream_kernel<8, true, 2, false> (bflow_corr_build_tiled: split8 arithmetic, fp32 tiled volume, D = 256 -- the product launch)"
LOOKUP_NAME = "corr_lookup_tile_kernel<float, 2, 256> (fused bezier, tiled planes, split out)"


def build(model, vox, cfg, low_params=None):
    dev = vox.device
    B, _, H, W = vox.shape
    out = []
    with torch.no_grad():
        grids, _ = model.gen_voxel_grids(vox)
        x5 = torch.cat(grids, dim=0)
        # (1) dominant kernel by time: the split-fp16 halo convolution.  Largest single launch: encoder layer1 conv, 64->64 3x3 on the
        #     5 stacked half-resolution maps (5B x 240 x 320), InstanceNorm statistics accumulated by the epilogue.
        #     Its input is produced the way the product produces it: the engine's own 7x7/2 stem + the fused normalise / ReLU kernel.
        n5, c0 = x5.shape[0], model.fnet_ev.conv1.out_channels
        h0, w0 = (x5.shape[2] - 1) // 2 + 1, (x5.shape[3] - 1) // 2 + 1
        st0 = torch.zeros((n5, c0, 2), dtype=torch.float64, device=dev)
        _, f0 = S.conv_stem(x5.contiguous(), S.PackedStemWeight().get(model.fnet_ev.conv1.weight), stats=st0, want_split=False, want_f32=True)
        cur, _ = S.norm_act(f0, (n5, h0, w0, c0), stats_a=st0, act_a=S.ACT_RELU)
        pk = S.PackedConvWeight().get(model.fnet_ev.layer1[0].conv1.weight)
        st = torch.zeros((8, n5, 64, 2), dtype=torch.float64, device=dev)      # 8 replicas, as the encoder uses them
        o32 = torch.empty((n5, 2, h0 * w0, 32), dtype=torch.float32, device=dev)
        out.append(dict(key="roofline", name=CONV_NAME, regex="conv_halo_kernel", bound="mfma",
                        launch=lambda: S.conv(cur, pk, stride=1, padding=1, want_split=False, out_f32=o32, stats=st),
                        flops=2.0 * n5 * h0 * w0 * 64 * 64 * 9,
                        # split input (4 B/elem) + fp32 output (4 B/elem) + packed weights
                        bytes=4.0 * n5 * h0 * w0 * 64 * 2 + 4.0 * 64 * 64 * 9))
        # (2) K5 correlation build on the same engine (HBM-write-bound by design): 393.2 MB algorithmic per sample
        D = model.fnet_ev.conv2.out_channels
        h8, w8 = H // 8, W // 8
        N = h8 * w8
        T = len(grids) - 1
        planes = model.fnet_ev.forward_split(x5, out_rows=hip.padded_rows(N)).planes
        # the launch of the product path (bflow_amd/corr.py): TILED planes, the model's default arithmetic ("split8": hi*hi on the fp16 rate +
        # both cross terms on the fp8 rate); the x8 operand planes come from one 3-us conversion launch in front of it (not timed here)
        prec = model.resolved_corr_precision()
        arith = {"split": hip.ARITH_SPLIT, "split8": hip.ARITH_SPLIT8}[prec]
        vol = torch.empty((T, B, N, hip.tiled_plane_size(h8, w8)), device=dev)
        p1, p2 = planes[:, :B], planes[:, B:]
        x8 = (hip.split_to_x8(p1), hip.split_to_x8(p2)) if arith == hip.ARITH_SPLIT8 else None
        out.append(dict(key="roofline_corr_build", name=K5_NAME if arith == hip.ARITH_SPLIT8 else K5_NAME.replace("2, false", "0, false").replace("split8", "split"),
                        regex="corr_stream_kernel", bound="hbm",
                        launch=lambda: hip.corr_build_tiled(p1, p2, vol, T, B, N, shared_f1=True, tiled_hw=(h8, w8), arithmetic=arith, x8=x8),
                        flops=2.0 * T * B * D * N * N, bytes=4.0 * ((1 + T) * B * D * N + T * B * N * N),
                        # second roof: matrix cores.  Matrix-pipe cost per fp32-class product: 3 fp16 units ("split") or
                        # 1 fp16 + 2 x 1/2 (fp8 runs at twice the fp16 rate) = 2 units ("split8")
                        mfma_peak=2500.0 / (2 if arith == hip.ARITH_SPLIT8 else 3)))
        # (3) the look-up gather (HBM-bound), 24.33 MB algorithmic per sample-iteration at C2: 100 taps read + 81 values written
        #     (4 B each; the split output is also 4 B per value) per (pixel, plane)
        cc = CorrComputation.from_packed(planes[:, :B], planes[:, B:], B, D, h8, w8, cfg["correlation"]["ev"]["levels"])
        cblk = CorrBlockParallelMultiTarget(corr_computation_events=cc, layout="tiled")   # the product path's layout
        params = (torch.randn(B, 2 * model.bezier_degree, h8, w8, device=dev) * 4 if low_params is None else low_params.clone())
        feat = cblk.new_output_split()
        coef = model._coefficients()
        out.append(dict(key="roofline_lookup", name=LOOKUP_NAME, regex="corr_lookup_tile_kernel", bound="hbm",
                        launch=lambda: cblk.lookup_bezier_split(params, coef, feat),
                        flops=None, bytes=4.0 * B * N * cblk.num_planes * (100 + 81), keep=(cblk, vol, planes, x8)))
    return out
