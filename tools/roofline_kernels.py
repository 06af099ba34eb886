"""The kernels bench.py prices against their rooflines, on the operands of the C2 workload (measurement harness only).

`build(model, vox, cfg)` returns a list of dicts {key, name, regex, bound, launch, flops, bytes}: `launch()` enqueues ONE launch of the
kernel on torch's current stream.  bench.py times them with hipEvents; tools/roofline_probe.py runs them under
`rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-include-regex <regex>` to produce profiles/r04_pmc.json.

  roofline                 the kernel with the largest TotalDurationNs in the C2-only kernel trace (profiles/r0x_rocprofv3_kernel_stats_c2only.csv):
                           conv_halo_kernel<2,3,3,...> on its largest launch, the feature encoder's layer-1 3x3 (64 -> 64 on 5 x 240 x 320)
  roofline_update_conv     the batch-1 small-grid 3x3 family on its largest launch: conv_halo8_pair_kernel<3,3>, the motion encoder's convc2 | convf2
                           (round 4 reported this one as `roofline`)
  roofline_corr_build      K5, the product launch (split8) ;  roofline_corr_build_split: the 3-pass fp32-class arithmetic on the same operands
  roofline_lookup          K7 at C2 (batch 1) ;  roofline_lookup_c4_shard: K7 on C4's per-GPU shard (batch 8)
  roofline_corr_build_c5   K5 at BASELINE configs[4]'s size (1024 x 1024, 5 event targets + 1 image target) with the arithmetic that config selects ("f16/w")
`frame_flops(model, ...)` counts the algorithmic FLOPs of one frame / one update iteration as the product executes them.
"""
import torch

from bflow_amd import hip, split as S
from bflow_amd.corr import CorrBlockParallelMultiTarget, CorrComputation

CONV_NAME = "conv_halo_stream_kernel<false> (encoder layer1 3x3 64->64, 5x240x320; rounds 1-4: conv_halo_kernel<2,3,3>)"
HALO8_NAME = ("conv_halo8_pair_kernel<3,3> (motion encoder convc2 | convf2 as ONE launch: 3x3 256->192 and 3x3 128->64, + bias + ReLU, 1x60x80 -- the 8-wave "
              "small-grid 3x3 kernel on its largest launch; the small-grid 3x3 / 1x5 / 5x1 family has the largest total time of the frame)")
HALO8_REGEX = "conv_halo8_pair_kernel"
K5_SPLIT_NAME = "corr_stream_kernel<8, true, 0, false> (bflow_corr_build_tiled: 3-pass split arithmetic = fp32 class, fp32 tiled volume, D = 256)"
LOOKUP_C4_NAME = "corr_lookup_tile_kernel<float, 2, 256, true> (C4 per-GPU shard: batch 8)"
K5_C5_NAME = "corr_stream_kernel (BASELINE configs[4]: 1024x1024, 5 event targets + 1 image target, f16/w = fp16 operands, one MFMA pass, fp32 tiled volume)"


def conv_flops(conv, Ho, Wo, images=1, cin=None):
    co, ci, kh, kw = conv.weight.shape
    return 2.0 * images * Ho * Wo * co * (ci if cin is None else cin) * kh * kw


def encoder_flops(enc, n, H, W):
    """BasicEncoder.forward_split on n images of H x W (extractor.py:103-125)."""
    h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    tot = conv_flops(enc.conv1, h, w, n)
    for li in (1, 2, 3):
        for blk in getattr(enc, f"layer{li}"):
            st = blk.conv1.stride[0]
            h2, w2 = (h - 1) // st + 1, (w - 1) // st + 1
            tot += conv_flops(blk.conv1, h2, w2, n) + conv_flops(blk.conv2, h2, w2, n)
            if blk.downsample is not None:
                tot += conv_flops(blk.downsample[0], h2, w2, n)
            h, w = h2, w2
    return tot + conv_flops(enc.conv2, h, w, n)


def frame_flops(model, B, H, W, iters):
    """Algorithmic (fp32-equivalent) FLOPs of ONE forward as the product executes it (raft.py:101-200 after the restructurings of
    DESIGN.md section 4): encoders; K5; per iteration the motion encoder, the six gate convolutions WITHOUT their loop-invariant context
    share (hoisted: computed once per frame, counted once), the Bezier head; the mask head once (last iteration only in test mode).
    Element-wise work, the look-up's interpolation and the up-sampling are not counted.  Returns (frame, one update iteration, parts)."""
    ub = model.update_block
    h, w = H // 8, W // 8
    N = h * w
    n_ev = (len(model.ev_corr_target_indices) + 1) if model.fnet_ev is not None else 0
    parts = {"fnet_ev": encoder_flops(model.fnet_ev, n_ev * B, H, W) if model.fnet_ev is not None else 0.0,
             "fnet_img": encoder_flops(model.fnet_img, 2 * B, H, W) if model.fnet_img is not None else 0.0,
             "cnet": encoder_flops(model.cnet, B, H, W)}
    D = (model.fnet_ev or model.fnet_img).conv2.out_channels
    T = len(model.lookup_timestamps)
    parts["corr_volume"] = 2.0 * T * B * D * N * N
    enc, gru, hd, cd = ub.encoder, ub.gru, ub.hidden_dim, ub.context_dim
    it = sum(conv_flops(c, h, w, B) for c in (enc.convc1, enc.convc2, enc.convf1, enc.convf2, enc.conv, ub.bezier_head.conv1, ub.bezier_head.conv2))
    gates = [getattr(gru, f"conv{g}{sfx}") for sfx in "12" for g in "zrq"]
    it += sum(conv_flops(c, h, w, B, cin=c.weight.shape[1] - cd) for c in gates)
    parts["update_iteration"] = it
    parts["hoisted_context_terms"] = sum(conv_flops(c, h, w, B, cin=cd) for c in gates)
    parts["mask_head"] = conv_flops(ub.mask[0], h, w, B) + conv_flops(ub.mask[2], h, w, B)
    frame = sum(v for k, v in parts.items() if k != "update_iteration") + iters * it
    return frame, it, parts
K5_NAME = "corr_stream_kernel<8, true, 2, false> (bflow_corr_build_tiled: split8 arithmetic, fp32 tiled volume, D = 256 -- the product launch)"
LOOKUP_NAME = "corr_lookup_tile_kernel<float, 2, 256, true> (fused bezier, tiled planes, split out; + the im2col rider of bflow_corr_lookup_im2col)"


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
        ub = model.update_block
        h8_, w8_ = H // 8, W // 8
        c1 = S.from_nchw(torch.relu(torch.randn(B, ub.encoder.convc1.out_channels, h8_, w8_, device=dev)))
        pk2 = ub._pk("convc2", lambda a=ub.encoder.convc2.weight: a)
        pkf2 = ub._pk("convf2", lambda a=ub.encoder.convf2.weight: a)
        f1_ = S.from_nchw(torch.relu(torch.randn(B, ub.encoder.convf1.out_channels, h8_, w8_, device=dev)))
        corbez = S.SplitTensor.empty(B, h8_, w8_, 256, dev)
        # the product launch (update.py step_split, one-queue form): bflow_conv_split_pair of the two 3x3s of the motion encoder
        out.append(dict(key="roofline_update_conv", name=HALO8_NAME, regex=HALO8_REGEX, bound="mfma",
                        launch=lambda: S.conv_pair(dict(x=c1, packed=pk2, padding=1, shift=ub.encoder.convc2.bias, act=S.ACT_RELU, out_split=corbez, channel_offset=0),
                                                   dict(x=f1_, packed=pkf2, padding=1, shift=ub.encoder.convf2.bias, act=S.ACT_RELU, out_split=corbez,
                                                        channel_offset=192)),
                        flops=conv_flops(ub.encoder.convc2, h8_, w8_, B) + conv_flops(ub.encoder.convf2, h8_, w8_, B),
                        bytes=4.0 * B * h8_ * w8_ * (256 + 192 + 128 + 64) + 4.0 * 9 * (192 * 256 + 64 * 128),
                        note="batch 1: 240 + 80 workgroups on 256 CUs, a link of a chain of dependent launches -- bound by the operand fill of a CU and launch "
                             "latency, not by the matrix cores"))
        out.insert(0, dict(key="roofline", name=CONV_NAME, regex="conv_halo_stream_kernel<false>", bound="mfma",
                        launch=lambda: S.conv(cur, pk, stride=1, padding=1, want_split=False, out_f32=o32, stats=st),
                        flops=2.0 * n5 * h0 * w0 * 64 * 64 * 9,
                        # split input (4 B/elem) + fp32 output (4 B/elem) + packed weights
                        bytes=4.0 * n5 * h0 * w0 * 64 * 2 + 4.0 * 64 * 64 * 9,
                        note="the kernel with the largest total time in the C2-only kernel trace; the frame's critical path (feature encoder)"))
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
        if arith != hip.ARITH_SPLIT:
            out.append(dict(key="roofline_corr_build_split", name=K5_SPLIT_NAME, regex="corr_stream_kernel", bound="hbm",
                            launch=lambda: hip.corr_build_tiled(p1, p2, vol, T, B, N, shared_f1=True, tiled_hw=(h8, w8), arithmetic=hip.ARITH_SPLIT),
                            flops=2.0 * T * B * D * N * N, bytes=4.0 * ((1 + T) * B * D * N + T * B * N * N), mfma_peak=2500.0 / 3,
                            note="the fp32-class number: three fp16 MFMA passes per product (what `value_split` runs)"))
        # (3) the look-up gather (HBM-bound), 24.33 MB algorithmic per sample-iteration at C2: 100 taps read + 81 values written
        #     (4 B each; the split output is also 4 B per value) per (pixel, plane)
        cc = CorrComputation.from_packed(planes[:, :B], planes[:, B:], B, D, h8, w8, cfg["correlation"]["ev"]["levels"])
        cblk = CorrBlockParallelMultiTarget(corr_computation_events=cc, layout="tiled")   # the product path's layout
        params = (torch.randn(B, 2 * model.bezier_degree, h8, w8, device=dev) * 4 if low_params is None else low_params.clone())
        feat = cblk.new_output_split()
        coef = model._coefficients()
        # the product launch at batch 1 carries the 7x7 windows of the Bezier parameters for convf1 as its first workgroups
        # (bflow_corr_lookup_im2col): their bytes (2 deg x 4 B read, ceil(49 x 2 deg / 32) x 32 x 4 B written per pixel) are part of the launch
        ck = (49 * 2 * model.bezier_degree + 31) // 32
        col = S.SplitTensor.empty(B, h8, w8, 49 * 2 * model.bezier_degree, dev)
        rider_bytes = 4.0 * B * N * (2 * model.bezier_degree + ck * 32)
        # 4 B N P (100 + 81) bytes -- the look-up alone.  The product launch at batch 1 also carries the im2col rider: timed
        # separately (`_lookup_with_rider`, folded into roofline_lookup.product_launch_with_rider by bench.py), its bytes never enter `frac`
        out.append(dict(key="roofline_lookup", name=LOOKUP_NAME.split(";")[0] + ")", regex="corr_lookup_tile_kernel", bound="hbm",
                        launch=lambda: cblk.lookup_bezier_split(params, coef, feat),
                        flops=None, bytes=4.0 * B * N * cblk.num_planes * (100 + 81), keep=(cblk, vol, planes, x8),
                        line_bytes=lookup_line_bytes(cblk, B)))
        out.append(dict(key="_lookup_with_rider", name=LOOKUP_NAME, regex="corr_lookup_tile_kernel", bound="hbm",
                        launch=lambda: cblk.lookup_bezier_split(params, coef, feat, im2col=(col, 7, 7, 3)),
                        flops=None, bytes=4.0 * B * N * cblk.num_planes * (100 + 81) + rider_bytes, rider_bytes=rider_bytes, keep=(cblk, col)))
    return out


def lookup_line_bytes(cblk, B):
    """The look-up's HBM floor at 128-B LINE granularity on tiled fp32 planes (a 4 x 8-element tile = one line): the 10 x 10 taps of a
    (pixel, plane) at a uniformly random offset touch E[rows] x E[cols] tiles -- rows: ceil((o + 10) / 4) over o = 0..3 -> 3.25 (capped by
    the plane's tile rows), cols: ceil((o + 10) / 8) over o = 0..7 -> 2.125 -- i.e. ~6.9 lines = 884 B for 400 B of taps at level 0, plus the
    81 x 4 B written.  The algorithmic `bytes` (181 x 4 B) cannot be met by ANY kernel on a materialised fp32 volume; this is what can."""
    h, w = cblk._hw
    N = h * w
    tot = 0.0
    for pl in cblk._planes:
        ph, pw = pl["hw"]
        rows = min(sum(-(-(o + 10) // 4) for o in range(4)) / 4.0, -(-ph // 4))
        cols = min(sum(-(-(o + 10) // 8) for o in range(8)) / 8.0, -(-pw // 8))
        tot += rows * cols * 128.0 + 81 * 4.0
    return B * N * tot


C4_CONV3_NAME = "conv_halo_kernel<2, 3, 3, false, false> (C4 per-GPU shard, batch 8: motion encoder convc2, 3x3 256->192 + bias + ReLU on 8x60x80)"
C4_STREAM_NIN_NAME = "conv_halo_stream_kernel<true> (C4 per-GPU shard, batch 8: feature encoder layer1 conv2, 3x3 64->64 normalise-on-load on 40x240x320, fp32 out + statistics)"
C4_GRU_NAME = "conv_halo_kernel<2, 1, 5, false, false> (C4 per-GPU shard, batch 8: SepConvGRU z|r, 1x5 [h | M] 288->256 on 8x60x80; plain epilogue here, the product launch carries the gate epilogue)"


def build_c4_convs(model, dev):
    """The three convolution kernels that lead the FULL kernel trace (profiles/r0x_rocprofv3_kernel_stats.csv: the default bench command, where
    the batch-8 workloads set whole-node throughput), each on its largest launch of the C4 per-GPU shard (batch 8).  Random operands of the
    product's shapes; durations do not depend on the values."""
    out = []
    g = torch.Generator(device="cpu").manual_seed(7)
    B, h8, w8 = 8, 60, 80
    ub = model.update_block
    with torch.no_grad():
        # (1) the large-grid per-item halo kernel, split output: convc2 of the motion encoder (update.py:90), 256 -> 192, bias + ReLU
        c1 = S.from_nchw(torch.relu(torch.randn((B, 256, h8, w8), generator=g)).to(dev))
        pk2 = S.PackedConvWeight().get(ub.encoder.convc2.weight)
        o2 = S.SplitTensor.empty(B, h8, w8, 192, dev)
        out.append(dict(key="roofline_conv3x3_c4", name=C4_CONV3_NAME, regex="conv_halo_kernel<2, 3, 3, false, false>", bound="mfma",
                        launch=lambda: S.conv(c1, pk2, padding=1, shift=ub.encoder.convc2.bias, act=S.ACT_RELU, out_split=o2),
                        flops=conv_flops(ub.encoder.convc2, h8, w8, B), bytes=4.0 * B * h8 * w8 * (256 + 192) + 4.0 * 9 * 192 * 256, keep=(c1, o2)))
        # (2) the persistent stream kernel, normalise-on-load: layer1.0.conv2 of the feature encoder on the shard's 5 x 8 = 40 half-resolution maps
        n, H2, W2 = 5 * B, 240, 320
        raw = torch.randn((n, 2, H2 * W2, 32), generator=g).to(dev)
        xs = torch.zeros((8, n, 64, 2), dtype=torch.float64, device=dev)
        xs[0, :, :, 0] = 0.0
        xs[0, :, :, 1] = float(H2 * W2)          # mean 0, variance 1 per (image, channel)
        pkc = S.PackedConvWeight().get(model.fnet_ev.layer1[0].conv2.weight)
        st = torch.zeros((8, n, 64, 2), dtype=torch.float64, device=dev)
        o32 = torch.empty((n, 2, H2 * W2, 32), dtype=torch.float32, device=dev)
        out.append(dict(key="roofline_conv_stream_nin_c4", name=C4_STREAM_NIN_NAME, regex="conv_halo_stream_kernel<true>", bound="mfma",
                        launch=lambda: S.conv_norm_in(raw, (n, H2, W2, 64), xs, pkc, stats=st, out_f32=o32),
                        flops=2.0 * n * H2 * W2 * 64 * 64 * 9, bytes=4.0 * n * H2 * W2 * 64 * 2 + 4.0 * 64 * 64 * 9, keep=(raw, o32, xs, st)))
        # (3) the 1x5 gate convolution of the separable GRU at batch 8: [h | M] = 128 + 160 channels -> z | r (256)
        hm = S.from_nchw(torch.randn((B, 288, h8, w8), generator=g).to(dev))
        wz = (torch.randn((256, 288, 1, 5), generator=g) / (288 * 5) ** 0.5).to(dev)
        pkg = S.PackedConvWeight().get(wz)
        og = S.SplitTensor.empty(B, h8, w8, 256, dev)
        out.append(dict(key="roofline_gru_conv_c4", name=C4_GRU_NAME, regex="conv_halo_kernel<2, 1, 5, false, false>", bound="mfma",
                        launch=lambda: S.conv(hm, pkg, padding=(0, 2), out_split=og),
                        flops=2.0 * B * h8 * w8 * 256 * 288 * 5, bytes=4.0 * B * h8 * w8 * (288 + 256) + 4.0 * 5 * 256 * 288, keep=(hm, og, wz)))
    return out


def build_big(model, cfg, dev):
    """The two launches that need their own (large) operands, built after the C2 set has been freed: K7 on C4's per-GPU shard (batch 8) and
    K5 at BASELINE configs[4]'s size with the arithmetic that config selects.  Operands are random features (the kernels' durations do
    not depend on the values: same bytes, same MFMA count), the Bezier parameters are N(0, 4 px)."""
    out = []
    g = torch.Generator(device="cpu").manual_seed(5)
    D, h8, w8, B = 256, 60, 80, 8
    N = h8 * w8
    levels = list(cfg["correlation"]["ev"]["levels"])
    T = len(levels)
    f1 = torch.randn((B, D, N), generator=g).to(dev)
    f2 = torch.randn((T * B, D, N), generator=g).to(dev)
    cc = CorrComputation.from_packed(hip.split_pack(f1), hip.split_pack(f2), B, D, h8, w8, levels)
    cblk = CorrBlockParallelMultiTarget(corr_computation_events=cc, layout="tiled")
    params = torch.randn(B, 2 * model.bezier_degree, h8, w8, device=dev) * 4
    feat = cblk.new_output_split()
    coef = model._coefficients()
    out.append(dict(key="roofline_lookup_c4_shard", name=LOOKUP_C4_NAME, regex="corr_lookup_tile_kernel", bound="hbm",
                    launch=lambda: cblk.lookup_bezier_split(params, coef, feat), flops=None,
                    bytes=4.0 * B * N * cblk.num_planes * (100 + 81), line_bytes=lookup_line_bytes(cblk, B), keep=(cblk, f1, f2)))
    # K5 at C5: 128 x 128 feature maps, the event group (shared reference map, 5 targets) and the image group (1 target) = two launches
    h5 = w5 = 128
    N5 = h5 * w5
    Np = hip.padded_rows(N5)
    ev1, ev2 = hip.split_pack(torch.randn((1, D, N5), generator=g).to(dev)), hip.split_pack(torch.randn((5, D, N5), generator=g).to(dev))
    im1, im2 = hip.split_pack(torch.randn((1, D, N5), generator=g).to(dev)), hip.split_pack(torch.randn((1, D, N5), generator=g).to(dev))
    vol5 = torch.empty((6, 1, N5, hip.tiled_plane_size(h5, w5)), device=dev)

    def k5_c5():
        hip.corr_build_tiled(ev1, ev2, vol5[:5], 5, 1, N5, shared_f1=True, tiled_hw=(h5, w5), arithmetic=hip.ARITH_F16)
        hip.corr_build_tiled(im1, im2, vol5[5:], 1, 1, N5, shared_f1=True, tiled_hw=(h5, w5), arithmetic=hip.ARITH_F16)
    out.append(dict(key="roofline_corr_build_c5", name=K5_C5_NAME, regex="corr_stream_kernel", bound="hbm", launch=k5_c5,
                    flops=2.0 * 6 * D * N5 * N5, bytes=2.0 * (6 + 2) * D * N5 + 4.0 * 6 * N5 * N5, mfma_peak=2500.0,
                    note="two launches (event group T = 5, image group T = 1) timed together; fp16 operands (2 B), fp32 volume (4 B)", keep=(vol5, Np)))
    return out
