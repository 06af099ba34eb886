"""Update block (K9-K11) -- drop-in for models/raft_spline/update.py:8-126 with the reference's parameter names.

Inference only.  `step_split()` is the product path: one GRU/Bezier iteration on the split-fp16 MFMA conv engine
(csrc/conv_split.hip) over blocked split workspaces:
  * no torch.cat: channel-block offsets and two-source convolutions ([h | M], [r*h | M]),
  * convz and convr are one 256-channel convolution; the loop-invariant context share of the six gate convolutions is
    evaluated once per frame and enters as an epilogue addend,
  * sigmoid / r*h / the GRU blend / `bezier += delta` and the re-emission of the Bezier channel block are conv epilogues,
  * the 7x7 Bezier convolution is an im2col + 1x1 GEMM; the look-up + correlation branch runs next to the Bezier branch,
  * the mask head only runs when its output is consumed (last iteration in test mode; raft.py:193-195).
`forward()` is the reference-shaped call (update.py:116-126): it builds a workspace from NCHW tensors and runs the SAME `step_split`,
so op-level tests against the oracle exercise the product path.  There is no library (MIOpen) variant in this package; the
configurations the engine cannot run raise `BflowHipError` (`check_engine_support`).
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import torch
import torch.nn as nn
from . import hip
from . import split as S
from .conv_train import Conv2d   # nn.Conv2d whose GPU training forward / backward run on the HIP conv engine


# head2 (Conv2d(256, 2*deg, 3)): the vector-ALU kernel takes 5 us per 4800 pixels, the MFMA halo kernel 17 us at 4800 pixels but only
# 40 us at 38400 (batch 8) where its grid fills the chip -- measured cross-over near 24000 pixels.
# Small-grid decisions (batch x h x w pixels), one constant each so that moving ONE cross-over for an A/B does not silently change the others.
# SMALL_GRID_MAX_PIXELS (batch 1-4 at 60 x 80) is where they all sit today:
SMALL_GRID_MAX_PIXELS = int(os.environ.get("BFLOW_SMALL_GRID_MAX_PIXELS", "20000"))     # (env: tools A/B, tools/c4_rank_probe.py)
THIN_HEAD_MAX_PIXELS = int(os.environ.get("BFLOW_THIN_HEAD_MAX_PIXELS", str(SMALL_GRID_MAX_PIXELS)))   # thin Bezier head vs the MFMA halo kernel (env: tools A/B)
ONE_QUEUE_MAX_PIXELS = SMALL_GRID_MAX_PIXELS     # the motion encoder's two branches as pair launches on one queue (else a side stream)
MASK_TILE96_MAX_PIXELS = SMALL_GRID_MAX_PIXELS   # 96-channel tiles for the mask head's 1x1 (one round of 228 workgroups at batch 1)
MERGE_BEZIER_BLOCK = os.environ.get("BFLOW_NO_MERGED_BEZIER") is None     # A/B switch (tools/): see SplitWorkspace
MASK_TILE = int(os.environ["BFLOW_MASK_TILE"]) if "BFLOW_MASK_TILE" in os.environ else None   # tools A/B: channel tile of the mask head's 1x1 (None: pick_tile)


class BezierHead(nn.Module):
    def __init__(self, bezier_degree: int, input_dim: int = 128, hidden_dim: int = 256):
        super().__init__()
        self.conv1 = Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = Conv2d(hidden_dim, bezier_degree * 2, 3, padding=1)


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim: int = 128, input_dim: int = 192 + 128):
        super().__init__()
        cin = hidden_dim + input_dim
        for sfx, k, p in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for gate in "zrq":
                setattr(self, f"conv{gate}{sfx}", Conv2d(cin, hidden_dim, k, padding=p))


class BasicMotionEncoder(nn.Module):
    def __init__(self, model_params: Dict[str, Any], output_dim: int = 128):
        super().__init__()
        cor_planes = self._num_cor_planes(model_params["correlation"], model_params["use_boundary_images"], model_params["use_events"])
        bezier_planes = model_params["bezier_degree"] * 2
        self.convc1 = Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = Conv2d(256, 192, 3, padding=1)
        self.convf1 = Conv2d(bezier_planes, 128, 7, padding=3)
        self.convf2 = Conv2d(128, 64, 3, padding=1)
        self.conv = Conv2d(64 + 192, output_dim - bezier_planes, 3, padding=1)

    @staticmethod
    def _num_cor_planes(corr_params: Dict[str, Any], use_boundary_images: bool, use_events: bool) -> int:
        """update.py:69-86."""
        assert use_events or use_boundary_images
        out = 0
        if use_events:
            ev = corr_params["ev"]
            assert len(ev["levels"]) > 0 and len(ev["levels"]) == len(ev["radius"])
            for lvl, rad in zip(ev["levels"], ev["radius"]):
                out += lvl * (2 * rad + 1) ** 2
        if use_boundary_images:
            out += corr_params["img"]["levels"] * (2 * corr_params["img"]["radius"] + 1) ** 2
        return out


class SplitLookup:
    """`corr` argument of `step_split` for the ordinary look-up (CorrBlockParallelMultiTarget.lookup_bezier_split into `out`): a callable the
    step invokes where the look-up belongs.  With im2col = (SplitTensor, kh, kw, padding) the same launch also writes the filter windows of the
    Bezier parameters (`im2col_rider`: the pyramid's look-up kernel can carry them)."""

    def __init__(self, corr_block, bezier: torch.Tensor, coef, out):
        self.corr_block, self.bezier, self.coef, self.out = corr_block, bezier, coef, out
        self.im2col_rider = bool(getattr(corr_block, "im2col_rider", False))
        self.after = None          # measurement hook (timers.StampTimer): called right behind the look-up launch

    def __call__(self, im2col=None):
        out = self.corr_block.lookup_bezier_split(self.bezier, self.coef, out=self.out, im2col=im2col)
        if self.after is not None:
            self.after()
        return out


class SplitWorkspace:
    """Buffers of one forward when the update block runs on the split-fp16 engine (blocked channels-last split tensors)."""

    def __init__(self, blk: "BasicUpdateBlock", batch: int, h: int, w: int, device):
        hd, md = blk.hidden_dim, blk.motion_dim
        # Layout of the GRU's motion input M.  merged: exactly cat([out, bezier]) of update.py:95-97 -- the 2*deg Bezier channels sit behind
        # the motion convolution's md - 2*deg channels INSIDE the last 32-channel block (md channels = md/32 k-blocks for the four gate
        # convolutions of an iteration instead of md/32 + 1: one k-block in nine less at DSEC size).  Needs producers that write a few channels
        # of a block and leave the rest alone: the thin head kernel (batch <= THIN_HEAD_MAX_PIXELS pixels) and a degree with 2*deg % 4 == 0.
        # Otherwise: [motion conv (md - 2deg, zero padded to md) | Bezier block (2*deg, zero padded to 32)].
        bz = blk.bezier_planes
        self.merged = MERGE_BEZIER_BLOCK and bz % 4 == 0 and batch * h * w <= min(THIN_HEAD_MAX_PIXELS, SMALL_GRID_MAX_PIXELS) and (md - bz) // 32 == (md - 1) // 32
        self.bez_channel = md - bz if self.merged else md                        # first Bezier channel of M
        self.H = S.SplitTensor.empty(batch, h, w, hd, device)                    # hidden state
        self.RH = S.SplitTensor.empty(batch, h, w, hd, device)                   # r * h
        self.Z = torch.empty((batch, hd // 32, h * w, 32), dtype=torch.float32, device=device)   # update gate z (blocked fp32)
        self.M = S.SplitTensor.empty(batch, h, w, md if self.merged else md + 32, device, zero=True)
        self.INP = None                                                          # relu(context) split, set by set_context
        self.corbez = S.SplitTensor.empty(batch, h, w, 256, device)
        self.COL = None                                                          # 7x7 windows of the Bezier parameters (one-queue path)
        self.inp_terms = None
        self.overlap = True                                                      # independent branches on a side stream


ITER_BRANCH = os.environ.get("BFLOW_NO_OVERLAP") is None    # tools A/B: the two motion-encoder branches on one queue
ONE_QUEUE = os.environ.get("BFLOW_NO_ONE_QUEUE") is None     # tools A/B: off = the side-stream form of the batch-1 motion encoder


class BasicUpdateBlock(nn.Module):
    def __init__(self, model_params: Dict[str, Any], hidden_dim: int = 128):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.motion_dim = model_params["motion"]["dim"]
        self.context_dim = model_params["context"]["dim"]
        self.bezier_planes = model_params["bezier_degree"] * 2
        self.encoder = BasicMotionEncoder(model_params, output_dim=self.motion_dim)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=self.context_dim + self.motion_dim)
        self.bezier_head = BezierHead(model_params["bezier_degree"], input_dim=hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(Conv2d(hidden_dim, 256, 3, padding=1), nn.ReLU(inplace=True), Conv2d(256, 64 * 9, 1, padding=0))

    def check_engine_support(self):
        """The split-fp16 engine works on 32-channel blocks; the Bezier block must fit one.  Anything else is refused loudly
        (there is no library fall-back)."""
        bad = []
        if self.hidden_dim % 32 or self.motion_dim % 32 or self.context_dim % 32:
            bad.append(f"hidden / motion / context dims must be multiples of 32 (got {self.hidden_dim}, {self.motion_dim}, {self.context_dim})")
        if self.bezier_planes > 32:
            bad.append(f"2 * bezier_degree = {self.bezier_planes} > 32 (the Bezier parameters travel as one 32-channel block)")
        if bad:
            raise hip.BflowHipError("BasicUpdateBlock: configuration not supported by the HIP conv engine: " + "; ".join(bad))

    # ------------------------------------------------------------------------------------------------ split-fp16 engine
    def _pk(self, name: str, weight_fn, cin_pad=None):
        cache = self.__dict__.setdefault("_pack_cache", {})
        if name not in cache:
            cache[name] = [S.PackedConvWeight(), None, None]
        entry = cache[name]
        srcs = weight_fn.__defaults__            # the source parameters the derived weight is built from
        key = tuple((t.data_ptr(), hip.tensor_version(t)) for t in srcs)
        if entry[1] != key:
            with torch.no_grad():
                entry[2] = weight_fn().contiguous()
            entry[1] = key
        return entry[0].get(entry[2], cin_pad)

    def _gate_weights(self, sfx: str, merged: bool = False):
        """Gate convolutions re-indexed for the engine's inputs.  The reference convolves cat([h, inp, motion]) (update.py:34-37);
        here the loop-invariant `inp` part is split off (it is convolved ONCE per frame and enters as an addend), and the
        remaining input is the virtual concatenation [h | motion conv (zero padded) | Bezier block (zero padded to 32)]."""
        g = self.gru
        hd, cd, md, bz = self.hidden_dim, self.context_dim, self.motion_dim, self.bezier_planes
        cz, cr, cq = getattr(g, "convz" + sfx), getattr(g, "convr" + sfx), getattr(g, "convq" + sfx)

        def hm(w):   # (Cout, hd+cd+md, kh, kw) -> (Cout, hd + md + 32, kh, kw); merged layout: (Cout, hd + md, kh, kw), the reference's own order
            co, _, kh, kw = w.shape
            if merged:
                return torch.cat([w[:, :hd], w[:, hd + cd:hd + cd + md]], dim=1)
            z = w.new_zeros
            mconv = w[:, hd + cd:hd + cd + md - bz]
            return torch.cat([w[:, :hd], mconv, z((co, bz, kh, kw)), w[:, hd + cd + md - bz:], z((co, 32 - bz, kh, kw))], dim=1)

        def zr_hm(a=cz.weight, b=cr.weight): return hm(torch.cat([a, b], dim=0))
        def q_hm(a=cq.weight): return hm(a)
        def zr_inp(a=cz.weight, b=cr.weight): return torch.cat([a, b], dim=0)[:, hd:hd + cd]
        def q_inp(a=cq.weight): return a[:, hd:hd + cd]
        def zr_bias(a=cz.bias, b=cr.bias): return torch.cat([a, b], dim=0)
        tag = sfx + ("m" if merged else "")
        return (self._pk("zr_hm" + tag, zr_hm), self._pk("q_hm" + tag, q_hm), self._pk("zr_inp" + sfx, zr_inp),
                self._pk("q_inp" + sfx, q_inp), zr_bias, cq.bias, cz.padding)

    def new_split_workspace(self, batch: int, h: int, w: int, device) -> SplitWorkspace:
        return SplitWorkspace(self, batch, h, w, device)

    def set_context_split(self, ws: SplitWorkspace, trunk: "S.SplitTensor", conv2: nn.Conv2d):
        """net = tanh(cnet[:, :hdim]) -> ws.H, inp = relu(cnet[:, hdim:]) -> ws.INP (raft.py:144-147): the encoder's 1x1 projection
        is applied as two convolutions with the activation in their epilogues.  Then the loop-invariant `inp` parts of the six
        gate convolutions are evaluated once (with the gate biases folded in)."""
        hd = self.hidden_dim

        def w_net(a=conv2.weight): return a[:hd]
        def w_inp(a=conv2.weight): return a[hd:]
        S.conv(trunk, self._pk("cnet_net", w_net), shift=conv2.bias[:hd].contiguous(), act=S.ACT_TANH, out_split=ws.H)
        ws.INP, _ = S.conv(trunk, self._pk("cnet_inp", w_inp), shift=conv2.bias[hd:].contiguous(), act=S.ACT_RELU)
        self._hoist_inp_terms(ws)

    def _hoist_inp_terms(self, ws: SplitWorkspace):
        """The `inp` (context) channels are the same in every iteration: their share of the six gate convolutions (+ the gate biases)
        is evaluated once per frame and enters the per-iteration convolutions as an epilogue addend (update.py:34-37)."""
        terms = []
        for sfx in ("1", "2"):
            _, _, zr_inp, q_inp, zr_bias, q_bias, pad = self._gate_weights(sfx)
            bz = self.__dict__.setdefault("_bias_cache", {})
            bkey = tuple((t.data_ptr(), hip.tensor_version(t), str(t.device)) for t in zr_bias.__defaults__)   # .to() / param swaps keep _version
            if ("zr" + sfx) not in bz or bz["zr" + sfx][0] != bkey:
                with torch.no_grad():
                    bz["zr" + sfx] = (bkey, zr_bias().contiguous())
            _, t_zr = S.conv(ws.INP, zr_inp, padding=pad, shift=bz["zr" + sfx][1], want_split=False, want_f32=True)
            _, t_q = S.conv(ws.INP, q_inp, padding=pad, shift=q_bias, want_split=False, want_f32=True)
            terms.append((t_zr, t_q))
        ws.inp_terms = terms

    def step_split(self, ws: SplitWorkspace, corr, bezier: torch.Tensor, need_mask: bool, mask_blocked: bool = False):
        """One iteration of update.py:116-126 on the split-fp16 engine.  corr: (B, P*81, h, w) fp32 (look-up output), the same as a blocked SplitTensor, a
        callable producing either (then the look-up itself overlaps with the Bezier branch), bezier: (B, 2*deg, h, w) fp32 updated IN PLACE.
        Returns the mask logits incl. bias (B, 576, h, w) fp32 -- mask_blocked: as the last convolution writes them, blocked fp32
        (B, 18, h*w, 32), for hip.cvx_upsample_blocked -- or None."""
        enc = self.encoder
        # ---- motion encoder (update.py:88-97); every bias + relu lives in a conv epilogue, every cat is a channel offset
        # The correlation branch (look-up -> 1x1 -> 3x3) and the Bezier branch (7x7 as im2col + 1x1 GEMM -> 3x3) are independent.
        # The LONGER one is issued on the side stream: the graph keeps the captured stream's nodes on one hardware queue, and a
        # cross-queue join costs ~10 us unless the other side finished long before (measured both ways).
        kh, kw = enc.convf1.kernel_size
        if ws.overlap and ONE_QUEUE and callable(corr) and getattr(corr, "im2col_rider", False) and ws.H.shape[0] * ws.H.H * ws.H.W <= ONE_QUEUE_MAX_PIXELS:
            # Small grids (batch 1 at DSEC size): the two branches as PAIR launches on ONE queue -- look-up | im2col, convc1 | convf1,
            # convc2 | convf2 -- instead of two queues: every cross-queue edge of the captured graph cost the chain 5-7 us
            # (profiles/r04_iteration_launches.txt), about what the overlap saved.
            if ws.COL is None:
                ws.COL = S.SplitTensor.empty(ws.H.shape[0], ws.H.H, ws.H.W, kh * kw * bezier.shape[1], bezier.device)
            cs = corr(im2col=(ws.COL, kh, kw, enc.convf1.padding))
            (c1, _), (f1, _), _ = S.conv_pair(
                dict(x=cs, packed=self._pk("convc1", lambda a=enc.convc1.weight: a), shift=enc.convc1.bias, act=S.ACT_RELU),
                dict(x=ws.COL, packed=self._pk("convf1_cols", lambda a=enc.convf1.weight: a.permute(0, 2, 3, 1).reshape(a.shape[0], -1, 1, 1)),
                     shift=enc.convf1.bias, act=S.ACT_RELU))
            S.conv_pair(
                dict(x=c1, packed=self._pk("convc2", lambda a=enc.convc2.weight: a), padding=1, shift=enc.convc2.bias, act=S.ACT_RELU,
                     out_split=ws.corbez, channel_offset=0),
                dict(x=f1, packed=self._pk("convf2", lambda a=enc.convf2.weight: a), padding=1, shift=enc.convf2.bias, act=S.ACT_RELU,
                     out_split=ws.corbez, channel_offset=192))
            return self._step_tail(ws, bezier, need_mask, mask_blocked)
        with hip.Branch(ws.overlap and ITER_BRANCH) as corr_branch:
            cs = corr() if callable(corr) else corr
            if not isinstance(cs, S.SplitTensor):
                cs = S.from_nchw(cs)
            c1, _ = S.conv(cs, self._pk("convc1", lambda a=enc.convc1.weight: a), shift=enc.convc1.bias, act=S.ACT_RELU)
            S.conv(c1, self._pk("convc2", lambda a=enc.convc2.weight: a), padding=1, shift=enc.convc2.bias, act=S.ACT_RELU,
                   out_split=ws.corbez, channel_offset=0)
        col = S.im2col_small(bezier, kh, kw, enc.convf1.padding)
        f1, _ = S.conv(col, self._pk("convf1_cols", lambda a=enc.convf1.weight: a.permute(0, 2, 3, 1).reshape(a.shape[0], -1, 1, 1)),
                       shift=enc.convf1.bias, act=S.ACT_RELU)
        S.conv(f1, self._pk("convf2", lambda a=enc.convf2.weight: a), padding=1, shift=enc.convf2.bias, act=S.ACT_RELU,
               out_split=ws.corbez, channel_offset=192)
        corr_branch.join()
        return self._step_tail(ws, bezier, need_mask, mask_blocked)

    def _step_tail(self, ws: SplitWorkspace, bezier: torch.Tensor, need_mask: bool, mask_blocked: bool = False):
        """The rest of the iteration behind the two motion-encoder branches: `conv`, the separable conv-GRU, the heads."""
        enc = self.encoder
        S.conv(ws.corbez, self._pk("conv", lambda a=enc.conv.weight: a), padding=1, shift=enc.conv.bias, act=S.ACT_RELU,
               out_split=ws.M, channel_offset=0, keep_pad=ws.merged)     # merged: the pad channels of the last block hold the Bezier parameters
        # ---- separable conv-GRU (update.py:33-48)
        for sfx, (t_zr, t_q) in zip(("1", "2"), ws.inp_terms):
            zr_hm, q_hm, _, _, _, _, pad = self._gate_weights(sfx, ws.merged)
            # z | r in one convolution; sigmoid, r * h (-> RH) and the final blend (-> H, in place) live in the conv epilogues
            S.conv(ws.H, zr_hm, x2=ws.M, padding=pad, addend=t_zr, gate=S.GATE_ZR, gate_h=ws.H, out_split=ws.RH, out_f32=ws.Z)
            S.conv(ws.RH, q_hm, x2=ws.M, padding=pad, addend=t_q, gate=S.GATE_BLEND, gate_h=ws.H, gate_z=ws.Z, out_split=ws.H)
        # ---- heads (update.py:17-18,111-114,120-125).  The mask head (needed on the last iteration only in test mode) and the
        # Bezier head both read the new hidden state and nothing of each other: two branches.
        mask = None
        with hip.Branch(ws.overlap and need_mask) as mask_branch:
            if need_mask:
                m1, _ = S.conv(ws.H, self._pk("mask0", lambda a=self.mask[0].weight: a), padding=1, shift=self.mask[0].bias, act=S.ACT_RELU)
                m2, m2f = S.conv(m1, self._pk("mask2", lambda a=self.mask[2].weight: a), shift=self.mask[2].bias, want_split=not mask_blocked,
                                 want_f32=mask_blocked,
                                 tile=MASK_TILE if MASK_TILE is not None else (96 if m1.shape[0] * m1.H * m1.W <= MASK_TILE96_MAX_PIXELS else None))
                # (batch 1: 96-channel tiles = 228 workgroups, one round, 229 KB of operands each, instead of 342 of 64 channels:
                #  3.643-3.647 vs 3.648-3.668 ms per frame over three alternating pairs)
                mask = m2f if mask_blocked else m2.to_nchw()     # blocked: the up-sampling kernel reads the convolution's own layout
        bh = self.bezier_head
        d1, _ = S.conv(ws.H, self._pk("head1", lambda a=bh.conv1.weight: a), padding=1, shift=bh.conv1.bias, act=S.ACT_RELU)
        # bezier += delta (bezier.py:137-139) and the new Bezier channel block of M are produced by the epilogue
        if d1.shape[0] * d1.H * d1.W <= THIN_HEAD_MAX_PIXELS:
            # (the merged layout of M needs a producer that writes a few channels of a block: SplitWorkspace.merged implies this branch)
            # a 2*deg-channel output on a grid of 40 patches: the thin vector-ALU kernel, not a 32-wide MFMA tile on 40 workgroups
            S.conv_thin_acc(d1, self.__dict__.setdefault("_head2_w", S.ThinConvWeight()).get(bh.conv2.weight), bh.conv2.bias, bezier,
                            out_split=ws.M, channel_offset=ws.bez_channel)
        else:
            assert not ws.merged, "SplitWorkspace.merged needs the thin head (THIN_HEAD_MAX_PIXELS below the workspace's pixel count?)"
            S.conv(d1, self._pk("head2", lambda a=bh.conv2.weight: a), padding=1, shift=bh.conv2.bias, acc_nchw=bezier, out_split=ws.M,
                   channel_offset=self.motion_dim)
        mask_branch.join()
        return mask

    def forward(self, net, inp, corr, bezier):
        """Reference-shaped call (update.py:116-126): (net, inp, corr, bezier) NCHW fp32 -> (net, mask, delta_bezier) without mutating
        the inputs.  Runs the product path: a split workspace is filled from the NCHW tensors and ONE `step_split` is executed."""
        self.check_engine_support()
        B, _, h, w = net.shape
        ws = self.new_split_workspace(B, h, w, net.device)
        ws.overlap = False
        ws.H = S.from_nchw(net.float().contiguous())
        ws.INP = S.from_nchw(inp.float().contiguous())
        self._hoist_inp_terms(ws)
        before = bezier.float().contiguous()
        after = before.clone()
        S.bezier_update(after, None, ws.M, ws.bez_channel // 32, channel_in_block=ws.bez_channel % 32)   # emit the Bezier channels of the GRU input
        mask = self.step_split(ws, corr.float().contiguous(), after, need_mask=True)
        return ws.H.to_nchw(), 0.25 * mask, after - before
