"""RAFTSpline -- MI355X-native drop-in for models/raft_spline/raft.py:14-200 (inference).

Same constructor (`RAFTSpline(config['model'])`), same `forward(voxel_grid, images, iters, flow_init, test_mode)`
signature and return types, same parameter names (`fnet_ev.*, fnet_img.*, cnet.*, update_block.*`) so the public
checkpoints load through `load_state_dict`.  What differs is how the hot path is executed:

    encoders            split-fp16 MFMA conv engine (bflow_conv_stem / bflow_conv_split + bflow_norm_act_split); the context
                        encoder runs on a side stream under the 5-image feature encoder                    (K4)
    correlation volume  bflow_corr_build_split on the encoder's own output layout (exact fp32 MFMA variant kept) (K5)
    pyramid             bflow_corr_pool2x2                                                                   (K6)
    per iteration       bflow_corr_lookup_bezier_split (Bezier evaluation + coords0 + 9x9 gather fused, written in the conv
                        engine's layout), 10 engine convolutions with bias / activation / concatenation / GRU gates /
                        parameter update in their epilogues (K7-K12); at batch 1 ten launches on one queue: the look-up
                        carries the im2col of the Bezier parameters (bflow_corr_lookup_im2col) and the motion encoder's
                        two branches are pair launches (bflow_conv_split_pair)
    last iteration      mask head + bflow_cvx_upsample_blocked (the mask in the convolution's own layout)    (K13)

The whole forward is free of host synchronisation, so it is captured once per input signature into a hipGraph
(bflow_amd/graph.py) and replayed: BY DEFAULT for the call val.py makes (eval(), inference_mode, test_mode=True; opt out with
BFLOW_HIPGRAPH=0 or enable_hipgraph(False)), for every inference forward after `enable_hipgraph()`; eager execution stays
available for debugging and stage timing.
Inputs must live on the GPU: there is no CPU fallback (the CPU restatement lives in oracle/ and is test-only).
"""
from __future__ import annotations

import os

from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import hip
from . import split as S
from .bezier import BezierCurves, polynomial_coefficients
from .corr import CorrBlockParallelMultiTarget, CorrComputation
from .extractor import BasicEncoder
from .timers import StageTimer
from .update import BasicUpdateBlock, SplitLookup



class EncodedFrame:
    """What the GRU loop reads of one frame (the output of RAFTSpline._encode): the correlation pyramid, the split workspace with the
    hidden state / context terms / initial Bezier block, the curve parameters and the look-up's output buffer."""

    def __init__(self, corr_block, ws, bezier, corr_feat):
        self.corr_block, self.ws, self.bezier, self.corr_feat = corr_block, ws, bezier, corr_feat

    def volume(self) -> torch.Tensor:
        return self.corr_block._pyramid[0][0]

    def small_tensors(self) -> List[torch.Tensor]:
        """Everything `_iterate` reads except the level-0 volume (which a double-buffered state receives in place, `volume_out`)."""
        ws = self.ws
        out = [t for t, _ in self.corr_block._pyramid[1:]]
        out += [ws.H.planes, ws.INP.planes, ws.M.planes, self.bezier]
        for t_zr, t_q in ws.inp_terms:
            out += [t_zr, t_q]
        return out


# layer of the event feature encoder behind which the context encoder is forked (0: at the start); BFLOW_CNET_FORK_LAYER for A/B
MASK_BLOCKED = os.environ.get("BFLOW_MASK_NCHW") is None     # the up-sampling reads the mask head's blocked fp32 output (no NCHW copy); env: tools A/B
CNET_FORK_LAYER = int(os.environ.get("BFLOW_CNET_FORK_LAYER", "0"))     # measured: 0 -> 272.8 / 272.3, 1 -> 271.9 / 271.3, 2 -> 263.2 / 263.6 frames/s


class RAFTSpline(nn.Module):
    def __init__(self, model_params: Dict[str, Any]):
        super().__init__()
        nbins_context = model_params["num_bins"]["context"]
        nbins_correlation = model_params["num_bins"]["correlation"]
        self.bezier_degree = model_params["bezier_degree"]
        self.detach_bezier = model_params["detach_bezier"]
        assert nbins_correlation is not None, "num_bins.correlation must be back-filled (modules/data_loading.py:63-68)"
        assert nbins_correlation > 0 and nbins_context > 0
        assert self.bezier_degree >= 1
        self.nbins_context = nbins_context
        self.nbins_corr = nbins_correlation

        corr_params = model_params["correlation"]
        self.corr_use_cosine_sim = corr_params["use_cosine_sim"]   # stored, never used (raft.py:33)
        ev = corr_params["ev"]
        self.ev_corr_target_indices = list(ev["target_indices"])
        self.ev_corr_levels = list(ev["levels"])
        self.ev_corr_radius = 4                                     # raft.py:38-40

        self.img_corr_params = None
        if model_params["use_boundary_images"]:
            self.img_corr_params = corr_params["img"]
            assert "levels" in self.img_corr_params and "radius" in self.img_corr_params

        self.hidden_dim = hdim = model_params["hidden"]["dim"]
        self.context_dim = cdim = model_params["context"]["dim"]
        cnorm = model_params["context"]["norm"]
        feature_dim = model_params["feature"]["dim"]
        fnorm = model_params["feature"]["norm"]

        context_dim = 0
        self.fnet_img = None
        if self.img_corr_params is not None:
            self.fnet_img = BasicEncoder(input_dim=3, output_dim=feature_dim, norm_fn=fnorm)
            context_dim += 3
        self.fnet_ev = None
        if model_params["use_events"]:
            assert 0 not in self.ev_corr_target_indices
            assert len(self.ev_corr_target_indices) > 0
            assert max(self.ev_corr_target_indices) < self.nbins_context
            assert len(self.ev_corr_target_indices) == len(self.ev_corr_levels)
            self.fnet_ev = BasicEncoder(input_dim=nbins_correlation, output_dim=feature_dim, norm_fn=fnorm)
            context_dim += nbins_context
        assert self.fnet_ev is not None or self.fnet_img is not None
        self.cnet = BasicEncoder(input_dim=context_dim, output_dim=hdim + cdim, norm_fn=cnorm)
        self.update_block = BasicUpdateBlock(model_params, hidden_dim=hdim)

        # look-up times are a property of the configuration (raft.py:156,170-177): build the Bezier coefficient table
        # ONCE instead of re-deriving it in numpy and copying it host->device every iteration (bezier.py:176-180)
        times: List[float] = []
        if self.fnet_ev is not None:
            dt = 1 / (self.nbins_context - 1)
            times += [dt * t for t in self.ev_corr_target_indices]
        if self.fnet_img is not None:
            times.append(1)
        self.lookup_timestamps = times
        self._coef = None
        # Correlation arithmetic (bflow_amd/corr.py PRECISIONS).  None = the model default: BFLOW_CORR_PRECISION when set, else "split8"
        # (hi*hi on the fp16 matrix rate, both cross terms on the fp8 rate, fp32 volume: 2.2e-5 px against the fp32 oracle at C2 / C5 where
        # the three-pass "split" measures 1.3e-5, K5 142 -> 95 us) for feature dims 128 / 256, "split" otherwise.
        # "f16" = fp16 operands AND fp16 volume (BASELINE configs[4]; 2.3-2.7e-3 px: outside the 1e-3 bar, opt-in).
        # An OPTIONAL model key `correlation.precision` (absent from the reference's YAML files, so they mean what they always meant)
        # presets it: configs.BASELINE_CONFIGS[4] ("fp16 MFMA correlation") selects "f16/w" this way.
        self.corr_precision: Optional[str] = corr_params.get("precision") if hasattr(corr_params, "get") else None
        # hipGraph replay of the inference forward.  "auto" (the default: what a caller gets who never heard of this package -- val.py ->
        # modules/raft_spline.py:57-58): eval() + grad disabled (val.py:75 inference_mode) + test_mode=True forwards replay a captured graph
        # and return PRIVATE copies of the outputs; "on" (enable_hipgraph()): every inference forward replays and returns the graph's
        # static buffers; "off" (enable_hipgraph(False) or BFLOW_HIPGRAPH=0): eager launches.
        self._graph_mode = "auto"
        self._graphs = None
        self._weights_gen = 0          # bumped by whatever replaces / rewrites parameters wholesale (graph.WeightsWatch reads it)
        self.stage_timer: Optional[StageTimer] = None
        self._probe = None            # tools only: callable(name) invoked at stage boundaries inside the captured forward

    def resolved_corr_precision(self) -> str:
        """The correlation arithmetic a forward will use (see `corr_precision`)."""
        from . import corr as _corr
        if self.corr_precision is not None:
            return _corr.check_precision(self.corr_precision)
        enc = self.fnet_ev if self.fnet_ev is not None else self.fnet_img
        D = enc.conv2.out_channels
        return _corr.default_precision(64 if D <= 64 else 128 if D <= 128 else 256, tiled=True)   # the ONE resolver (BFLOW_CORR_PRECISION, else split8 / split) on the PADDED dim

    # ---------------------------------------------------------------------------------------- reference API
    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def initialize_flow(self, input_: torch.Tensor):
        """raft.py:80-86 (coords0 is implicit in the fused look-up; materialised here only for API parity)."""
        N, _, H, W = input_.shape
        ys, xs = torch.meshgrid(torch.arange(H // 8, device=input_.device), torch.arange(W // 8, device=input_.device), indexing="ij")
        coords0 = torch.stack([xs, ys], dim=0).float()[None].repeat(N, 1, 1, 1)
        return coords0, BezierCurves.create_from_voxel_grid(input_, downsample_factor=8, bezier_degree=self.bezier_degree)

    def gen_voxel_grids(self, input_: torch.Tensor):
        """raft.py:88-99: channel windows [idx, idx+nbins_corr) for idx in {0} + target indices; context = last bins."""
        assert self.nbins_context + self.nbins_corr - 1 == input_.shape[-3]
        grids = [input_[:, idx:idx + self.nbins_corr, ...] for idx in [0] + self.ev_corr_target_indices]
        return grids, input_[:, -self.nbins_context:, ...]

    # ---------------------------------------------------------------------------------------- execution control
    def enable_hipgraph(self, enabled: Optional[bool] = True):
        """Replay EVERY inference forward from a captured hipGraph (one graph per input signature), or none (`enabled=False`);
        `enabled=None` returns to the default "auto" mode.
        Without this call the model is in "auto" mode: forwards in eval() with grad disabled and test_mode=True -- the call
        modules/raft_spline.py:57-58 makes under val.py:75 -- replay by themselves and return private copies of their outputs.
        ALIASING (this explicit mode only): the returned BezierCurves wrap the graph's static output buffers; the next forward with the
        same signature overwrites them.  Clone (`curves.detach(clone=True)`) whatever must outlive the next call -- Validator does."""
        from .graph import GraphCache
        if enabled is None:
            self._graph_mode = "auto"
            return self
        self._graph_mode = "on" if enabled else "off"
        if not enabled:
            self._graphs = None
        elif self._graphs is None:
            self._graphs = GraphCache(self)
        return self

    def graph_replays(self) -> int:
        """Number of forwards served by a hipGraph replay so far (0 = every forward ran eager)."""
        return 0 if self._graphs is None else self._graphs.replays

    def _use_graph(self, test_mode: bool) -> bool:
        if self.stage_timer is not None:                     # stage timing brackets eager launches with hipEvents
            return False
        if self._graph_mode == "on":
            return True
        if self._graph_mode == "off" or not test_mode or torch.is_grad_enabled():
            return False
        if os.environ.get("BFLOW_HIPGRAPH", "1").lower() in ("0", "off", "false", "no"):
            return False
        if self._graphs is None:
            from .graph import GraphCache
            self._graphs = GraphCache(self)
        return True

    # whatever rewrites the weights wholesale tells the graph caches in O(1) (graph.WeightsWatch); in-place edits of single tensors are
    # caught by its version-counter sum
    def _apply(self, fn, *args, **kwargs):
        self._weights_gen += 1
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._weights_gen += 1
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode: bool = True):
        if mode:
            self._weights_gen += 1                           # a training step rewrites every parameter
        return super().train(mode)

    def enable_stage_timing(self, enabled: bool = True):
        """hipEvent stage timers with the reference's hook names (raft.py:116-186, utils/timers.py); eager mode only."""
        self.stage_timer = StageTimer() if enabled else None
        return self

    def _coefficients(self) -> np.ndarray:
        if self._coef is None:
            self._coef = polynomial_coefficients(self.lookup_timestamps, self.bezier_degree)
        return self._coef

    def forward(self,
                voxel_grid: Optional[torch.Tensor] = None,
                images: Optional[List[torch.Tensor]] = None,
                iters: int = 12,
                flow_init: Optional[BezierCurves] = None,
                test_mode: bool = False):
        assert voxel_grid is not None or images is not None
        assert iters > 0
        ref = voxel_grid if voxel_grid is not None else images[0]
        if not ref.is_cuda:
            raise hip.BflowHipError("RAFTSpline (bflow_amd) runs on MI355X only: move the inputs and the module to the GPU "
                                    "(the CPU restatement lives in oracle/ and is test infrastructure)")
        if self.training:
            if not torch.is_grad_enabled():
                raise hip.BflowHipError("RAFTSpline.forward in train() mode under no_grad / inference_mode: the differentiable forward would run "
                                        "every convolution outside the conv engine's autograd path.  Call model.eval() for inference (BatchNorm "
                                        "running statistics, as val.py does) or enable grad for a training step")
            # differentiable forward (HIP forward/backward for K5-K7/K13, torch autograd for the convolutions)
            from .training import forward_train
            with torch.cuda.device(ref.device):
                low, ups = forward_train(self, voxel_grid, images, iters, None if flow_init is None else flow_init.get_params())
            return (BezierCurves(low), ups[-1]) if test_mode else ups
        # kernels are launched on the CURRENT device's stream: make the inputs' device current for the duration of the call
        use_graph = self._use_graph(test_mode)      # (asks whether the CALLER disabled grad: before this method's own no_grad)
        with torch.cuda.device(ref.device), torch.no_grad():
            init = None if flow_init is None else flow_init.get_params()
            if use_graph:
                low, ups = self._graphs.run(voxel_grid, images, iters, init, test_mode)
                if self._graph_mode == "auto":
                    # the drop-in seam hands out tensors the caller owns: nothing aliases the graph's static buffers (two small
                    # device-to-device copies, 77 KB + 4.9 MB at DSEC size, enqueued behind the replay)
                    low, ups = low.clone(), [ups[-1].clone()]
            else:
                low, ups = self._forward_impl(voxel_grid, images, iters, init, test_mode)
        if test_mode:
            return BezierCurves(low), BezierCurves(ups[-1])
        return [BezierCurves(u) for u in ups]

    def check_engine_support(self):
        """Inference runs on the hand-written HIP engine ONLY.  A configuration it cannot run is refused here, loudly, instead of being
        routed to a library: there is no MIOpen / rocBLAS / CPU path in this package's inference forward."""
        bad = []
        for name in ("fnet_ev", "fnet_img", "cnet"):
            net = getattr(self, name)
            if net is None:
                continue
            if net.norm_fn not in ("instance", "batch", "group", "none"):
                bad.append(f"{name}.norm_fn = {net.norm_fn!r}")
            if name != "cnet" and net.conv2.out_channels > 256:
                bad.append(f"{name} output dim {net.conv2.out_channels} (the correlation kernels contract at most 256 feature channels; smaller dims are zero-padded to 64 / 128 / 256)")
        if self.fnet_ev is not None and len(self.ev_corr_target_indices) + 1 > 8:
            bad.append(f"{len(self.ev_corr_target_indices)} event targets (the stem reads at most 8 channel windows in place)")
        if bad:
            raise hip.BflowHipError("RAFTSpline: configuration not supported by the HIP engine: " + "; ".join(bad))
        self.update_block.check_engine_support()

    # ---------------------------------------------------------------------------------------- the hot path
    def _forward_impl(self, voxel_grid, images, iters: int, flow_init: Optional[torch.Tensor], test_mode: bool):
        """raft.py:101-200 on the HIP kernels.  No host<->device synchronisation anywhere (hipGraph-capturable).
        Two phases: `_encode` (raft.py:118-162: encoders, correlation volume + pyramid, context split, initial curve) produces everything
        the GRU loop reads, `_iterate` (raft.py:164-195) runs the loop.  bflow_amd/pipeline.py overlaps the two phases of consecutive
        frames."""
        return self._iterate(self._encode(voxel_grid, images, flow_init), iters, test_mode)

    def _encode(self, voxel_grid, images, flow_init: Optional[torch.Tensor], volume_out: Optional[torch.Tensor] = None) -> "EncodedFrame":
        hdim, cdim = self.hidden_dim, self.context_dim
        tm = self.stage_timer
        pr = self._probe
        corr_ev = corr_img = None
        context_input = None

        self.check_engine_support()

        def encode_pair(net, x, n_ref, levels, after_layer=None):
            """Feature encoder on a stacked batch [reference | targets] -> CorrComputation.  The last convolution writes K5's operand
            format directly: (2, nb, D/32, Np, 32), tail rows zero."""
            nb, _, Hh, Ww = x.shape
            h8, w8 = Hh // 8, Ww // 8
            D = net.conv2.out_channels
            Dp = 64 if D <= 64 else 128 if D <= 128 else 256
            if Dp != D:
                # K5's streaming kernel contracts 64, 128 or 256 channels: other feature dims are ZERO-PADDED to the next of them (whole
                # channel blocks of zeros add nothing to a dot product).  The kernel then divides by sqrt(Dp) where corr.py:270 divides by
                # sqrt(D): both feature maps leave the encoder's projection multiplied by (Dp / D)^(1/4), so that the product carries the
                # missing sqrt(Dp / D) (an fp32 rounding of each feature apart -- not bit-faithful to the reference's order, inside the parity bar)
                feat_out.pop(id(net), None)
                buf = S.SplitTensor.empty(nb, h8, w8, Dp, x.device, rows=hip.padded_rows(h8 * w8), zero=True)
                planes = net.forward_split(x, out=buf, out_gain=float((Dp / D) ** 0.25), after_layer=after_layer).planes
                return CorrComputation.from_packed(planes[:, :n_ref], planes[:, n_ref:], n_ref, Dp, h8, w8, levels)
            pre = feat_out.pop(id(net), None)        # (output tensor with zeroed pad rows, event): allocated on the context branch
            if pre is not None and after_layer is None:
                planes = net.forward_split(x, out=pre[0], out_ready=pre[1]).planes
            else:
                planes = net.forward_split(x, out_rows=hip.padded_rows(h8 * w8), after_layer=after_layer).planes
            return CorrComputation.from_packed(planes[:, :n_ref], planes[:, n_ref:], n_ref, D, h8, w8, levels)

        # ---- inputs of the three encoders (raft.py:118-141)
        grids = None
        if self.fnet_ev is not None:
            assert voxel_grid is not None
            voxel_grid = voxel_grid.contiguous().float()
            grids, context_input = self.gen_voxel_grids(voxel_grid)
        img_in = ctx_general = None
        if self.fnet_img is not None:
            assert images is not None and len(images) == 2
            # raft.py:134-140 without a torch launch: the normalisation 2 * (x / 255) - 1, the stacking of the two images (extractor.py:106-110)
            # and cat((context_grid, img0)) all happen in the stem kernel's load (S.StemInput); uint8 / fp32 images are read as they are
            images = [x.contiguous() if x.dtype in (torch.uint8, torch.float32) else x.float().contiguous() for x in images]
            if images[0].dtype != images[1].dtype:      # the reference calls .float() on each image (raft.py:134): a mixed pair is legal
                images = [x.float() for x in images]
            if images[0].shape != images[1].shape or images[0].dim() != 4 or images[0].shape[1] != 3:
                raise hip.BflowHipError(f"images: two (B, 3, H, W) tensors of equal shape expected, got {tuple(images[0].shape)} and {tuple(images[1].shape)}")
            img_in = S.StemInput([(images[0], 0), (images[1], 0)], 3, norm=True)
            if context_input is None:
                ctx_general = S.StemInput([(images[0], 0)], 3, norm=True)
            else:
                ctx_general = S.StemInput([(voxel_grid, voxel_grid.shape[1] - self.nbins_context)], self.nbins_context, extra=images[0], extra_norm=True)
            context_input = ctx_general          # (only its shape / device are read below)
        assert context_input is not None
        B, _, H, W = context_input.shape
        assert H % 8 == 0 and W % 8 == 0                                       # bezier.py:67-68
        h, w = H // 8, W // 8
        device = context_input.device

        # ---- context encoder on a side stream: a batch-1 chain of small launches that hides under the 5-image feature encoder.  It is
        # can be forked BEHIND layer `CNET_FORK_LAYER` of the feature encoder (0 = at the start, the default): starting it later, next to the
        # smaller stages and K5 only, conserves the total (the work is the same and the chip is busy either way) -- measured, no gain
        ub = self.update_block
        state = {}
        feat_out = {}

        def run_cnet():
            if tm: tm.start("cnet")
            with hip.Branch(tm is None) as br:
                if pr: pr("cnet.begin")
                # the feature encoders' outputs (K5's operand format: pad rows zero) are allocated HERE: their strided pad-row fill leaves the
                # feature encoder's chain (7.5 us right in front of its last convolution) for the start of the context branch
                if br.enabled and os.environ.get("BFLOW_NO_FEAT_PREALLOC") is None:      # (A/B switch, tools/)
                    for net_, nimg in ((self.fnet_ev, (1 + len(self.ev_corr_target_indices)) * B if self.fnet_ev is not None else 0),
                                       (self.fnet_img, 2 * B)):
                        if net_ is not None:
                            t_ = S.SplitTensor.empty(nimg, h, w, net_.conv2.out_channels, device, rows=hip.padded_rows(h * w), zero_tail=True)
                            ev_ = torch.cuda.Event()
                            ev_.record()
                            feat_out[id(net_)] = (t_, ev_)
                ws_ = ub.new_split_workspace(B, h, w, device)
                state["bezier0"] = torch.zeros((B, 2 * self.bezier_degree, h, w), dtype=torch.float32, device=device)   # raft.py:150
                ws_.overlap = tm is None and hip.BRANCHING
                ctx_in = context_input
                if self.fnet_img is None:      # the context bins are the LAST channels of the voxel grid: one window, read in place
                    ctx_in = S.ChannelWindows(voxel_grid, [voxel_grid.shape[1] - self.nbins_context], self.nbins_context)
                ub.set_context_split(ws_, self.cnet.forward_split(ctx_in, trunk_only=True), self.cnet.conv2)
                if pr: pr("cnet.end")
            if tm: tm.stop("cnet")
            state["ws"], state["branch"] = ws_, br

        fork_layer = CNET_FORK_LAYER if (self.fnet_ev is not None and tm is None) else 0
        if fork_layer == 0:
            run_cnet()

        # ---- feature encoders + correlation volumes
        if pr: pr("fnet.begin")
        if self.fnet_ev is not None:
            if tm: tm.start("fnet_ev")
            # [reference | targets] = channel windows of the voxel grid, read in place by the stem kernel (no torch.cat)
            stacked = S.ChannelWindows(voxel_grid, [0] + list(self.ev_corr_target_indices), self.nbins_corr)
            corr_ev = encode_pair(self.fnet_ev, stacked, B, self.ev_corr_levels, after_layer=(fork_layer, run_cnet) if fork_layer else None)
            if tm: tm.stop("fnet_ev")
        if self.fnet_img is not None:
            if tm: tm.start("fnet_img")
            corr_img = encode_pair(self.fnet_img, img_in, B, self.img_corr_params["levels"])
            if tm: tm.stop("fnet_img")

        if flow_init is not None:
            bezier = torch.zeros((B, 2 * self.bezier_degree, h, w), dtype=torch.float32, device=device)   # raft.py:150
            assert flow_init.shape == bezier.shape
            bezier += flow_init                                               # raft.py:152-153
        else:
            bezier = state["bezier0"]                                          # zeros, filled on the context branch (off the critical chain)

        if pr: pr("fnet.end")
        if tm: tm.start("corr computation")
        corr_block = CorrBlockParallelMultiTarget(corr_computation_events=corr_ev, corr_computation_frames=corr_img, layout="tiled",
                                                  precision=self.resolved_corr_precision(), volume_out=volume_out)
        if tm: tm.stop("corr computation")
        if pr: pr("corr.end")
        ws = state["ws"]
        state["branch"].join()
        if pr: pr("joined")

        corr_feat = corr_block.new_output_split()
        if flow_init is not None:     # (zero parameters: their channels of M are the zeros the workspace was allocated with -- no launch)
            S.bezier_update(bezier, None, ws.M, ws.bez_channel // 32, channel_in_block=ws.bez_channel % 32)     # emit the initial Bezier channels
        return EncodedFrame(corr_block, ws, bezier, corr_feat)

    def _iterate(self, fr: "EncodedFrame", iters: int, test_mode: bool):
        tm = self.stage_timer
        pr = self._probe
        ub = self.update_block
        corr_block, ws, bezier, corr_feat = fr.corr_block, fr.ws, fr.bezier, fr.corr_feat
        coef = self._coefficients()
        ups: List[torch.Tensor] = []
        if tm: tm.start("all iters")
        for itr in range(iters):
            need_mask = (not test_mode) or itr == iters - 1
            if pr: pr(f"iter{itr}.begin")
            if tm is None:
                # the look-up runs inside the step, next to the (independent) Bezier branch of the motion encoder
                look = SplitLookup(corr_block, bezier, coef, corr_feat)
                if pr:
                    look.after = lambda k=itr: pr(f"iter{k}.lookup_end")
                mask = ub.step_split(ws, look, bezier, need_mask, mask_blocked=MASK_BLOCKED)
            else:
                # stage timing (eager): the same kernels, the look-up timed on its own, no side-stream overlap
                tm.start("1 iter")
                tm.start("corr lookup (per iter)")          # includes 'get_flow (per iter)': fused into the gather
                corr_block.lookup_bezier_split(bezier, coef, out=corr_feat)
                tm.stop("corr lookup (per iter)")
                tm.start("update (per iter)")
                mask = ub.step_split(ws, corr_feat, bezier, need_mask)
                tm.stop("update (per iter)")
                tm.stop("1 iter")
            if need_mask:
                ups.append(hip.cvx_upsample_blocked(bezier, mask, 0.25) if (tm is None and MASK_BLOCKED) else hip.cvx_upsample(bezier, mask, None, 0.25))
        if tm: tm.stop("all iters")
        if pr: pr("iters.end")
        return bezier, ups
