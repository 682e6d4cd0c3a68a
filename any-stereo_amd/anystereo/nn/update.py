"""Iterative update operator — HIP-backed mirror of models/*/update.py:16-136.

Module / parameter names equal the reference's (`update_block.gru04.convz.weight`,
`update_block.encoder.convc1.bias`, ...) so checkpoints load unchanged; `forward` signatures are
the reference's.  The nn.Conv2d children only *hold* the parameters: the arithmetic runs in
libanystereo_hip.so (fp32-MFMA implicit GEMM, conv.hip):

  * ConvGRU: convz‖convr run as ONE conv (Cout = 2*hidden) over the un-materialised concat
    [h, x...], the epilogue emits z and r⊙h; convq consumes [r⊙h, x...] and its epilogue emits
    (1-z)·h + z·tanh(·)   — 2 launches instead of 3 convs + 2 cats + 6 pointwise kernels.
  * BasicMotionEncoder: convc2 / convd2 write into the two halves of one buffer (no cat),
    the 7x7 1-channel conv and the 256->1 head conv use direct kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import grad as G
from .. import ops
from ..harness.timing import scope


def _f(t):
    return t.float().contiguous()


def _links() -> bool:
    """Conv -> conv links as blocked split-fp16 tensors (ops.BS8: bit-identical results, a quarter of the consumer's patch
    loads): only the split-precision kernel reads / writes them.  ANYSTEREO_BS_LINKS=0 keeps fp32 links (A/B timing)."""
    return _LINKS_ENV and ops.get_precision() == "split"


_LINKS_ENV = __import__("os").environ.get("ANYSTEREO_BS_LINKS", "1") != "0"


def _twin(t):
    """The blocked twin a producer attached to its fp32 result (same values), or the tensor itself.  The twin is used only
    while the fp32 tensor is unmodified since it was attached (same version counter): an in-place update of a hidden state
    by the caller silently falls back to the fp32 tensor."""
    ent = getattr(t, "_as_bs", None)
    if ent is None or not _links():
        return t
    bs, ver = ent
    return bs if (ver == t._version and tuple(bs.shape) == tuple(t.shape)) else t


def _attach_twin(t, bs):
    t._as_bs = (bs, t._version)


# Training (gradients required): the reference's formulation op for op (gates as separate pointwise ops under autograd);
# the 3x3 / 1x1 convolutions run forward and dgrad on the implicit-GEMM kernel (grad.Conv2dSame; convz and convr as ONE
# conv over concatenated weights), wgrad on the library.  The 7x7 one-channel conv and the 256 -> 1 head conv stay nn.Conv2d.
# The fused GRU epilogues are the inference path; their backward is the next step.
_train = G.needs_grad
_cs = G.conv2d_same


class DispHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256, output_dim=1):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, output_dim, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self._p1 = ops.PackedConv()
        self._p2 = ops.PackedConv()

    def forward(self, x, addend=None):
        """delta = conv2(relu(conv1(x))) (update.py:23-24); with `addend` the result is addend + delta (the loop's
        `disp = disp + delta_disp`, fused into the last kernel)."""
        if _train(x, self.conv1.weight):
            hid = _cs(self, "conv1", x, self.conv1.weight, self.conv1.bias, relu=True)  # update.py:23-24
            out = _cs(self, "conv2", hid, self.conv2.weight, self.conv2.bias) if self.train_conv2_hip else self.conv2(hid)
            return out if addend is None else addend + out
        links = _links()
        if self.fused_head and self.conv2.out_channels == 1 and ops.get_precision() == "split":
            # conv2 (3x3, 256 -> 1) folded into conv1's epilogue: per 64-channel tile the nine per-tap channel reductions of
            # relu(conv1) — the 256-channel hidden layer is never written — then the shifted 9-tap sum over the four tiles
            with scope("disp_head_conv1"):
                taps = ops.conv2d([_twin(_f(x))], self._p1.get([self.conv1.weight], [self.conv1.bias]), act=L.ACT_RELU,
                                  epilogue=L.EPI_RELU_TAPS, tap_w=self._tap_weights())
            ops.mark_fine("head_conv1_end")
            with scope("disp_head_conv2"):
                return ops.tap_shift_sum(taps, _f(self.conv2.bias.detach()), None if addend is None else _f(addend))
        with scope("disp_head_conv1"):
            if links:  # hidden layer handed to conv2 as a blocked tensor only
                b, _, hh, ww = x.shape
                t = ops.BS8.empty(b, self.conv1.out_channels, hh, ww, x.device)
                ops.conv2d([_twin(_f(x))], self._p1.get([self.conv1.weight], [self.conv1.bias]), act=L.ACT_RELU, out_bs=t, bs_only=True)
            else:
                t = ops.conv2d([_f(x)], self._p1.get([self.conv1.weight], [self.conv1.bias]), act=L.ACT_RELU)
        if self.conv2.out_channels == 1:
            # 3x3, 256 -> 1 as (1x1, 256 -> 9 tap planes on the MFMA path) + a 9-tap shifted sum
            with scope("disp_head_conv2"):
                taps = ops.conv2d([t], self._p2.get([self.conv2.weight], [None],
                                                    transform=lambda w: w[0].permute(1, 2, 0).reshape(9, -1, 1, 1).contiguous()))
                return ops.tap_shift_sum(taps, _f(self.conv2.bias.detach()), None if addend is None else _f(addend))
        out = ops.conv2d([t], self._p2.get([self.conv2.weight], [self.conv2.bias]))
        return out if addend is None else addend + out


    fused_head = __import__("os").environ.get("ANYSTEREO_FUSED_HEAD", "1") != "0"
    # training: conv2 (256 -> 1) forward / dgrad / wgrad on this library's kernels like every other layer of the update block
    # (ANYSTEREO_TRAIN_HEAD_CONV2=0: MIOpen, 16 forward + 16 backward library calls per step)
    train_conv2_hip = __import__("os").environ.get("ANYSTEREO_TRAIN_HEAD_CONV2", "1") != "0"

    def taps_ok(self, x) -> bool:
        return (self.fused_head and self.conv2.out_channels == 1 and ops.get_precision() == "split" and not _train(x, self.conv1.weight))

    def taps(self, x):
        """First half of forward(): the per-tap channel reductions of conv2 over relu(conv1(x)) ([B, 4*9, H, W]); finish with
        `finish(taps, addend)` or hand them to the fused loop front (BasicMotionEncoder.forward_front)."""
        with scope("disp_head_conv1"):
            return ops.conv2d([_twin(_f(x))], self._p1.get([self.conv1.weight], [self.conv1.bias]), act=L.ACT_RELU,
                              epilogue=L.EPI_RELU_TAPS, tap_w=self._tap_weights())

    def finish(self, taps, addend=None):
        with scope("disp_head_conv2"):
            return ops.tap_shift_sum(taps, _f(self.conv2.bias.detach()), None if addend is None else _f(addend))

    def _tap_weights(self):
        """conv2.weight [1, C, 3, 3] as [C, 9] (tap = ky*3+kx), cached per weight version."""
        w = self.conv2.weight
        key = (w.data_ptr(), w._version, w.device)
        ent = self.__dict__.get("_tapw")
        if ent is None or ent[0] != key:
            ent = (key, w.detach()[0].reshape(w.shape[1], 9).float().contiguous())
            self.__dict__["_tapw"] = ent
        return ent[1]


class FlowHead(DispHead):
    def __init__(self, input_dim=128, hidden_dim=256, output_dim=2):
        super().__init__(input_dim, hidden_dim, output_dim)


def _context_window(cz, cr, cq):
    """cz, cr, cq are normally the three 128-channel views of ONE context tensor
    (`conv(i).split(...)`, continuous_IGEVstereo.py:273); find that tensor so the kernels index it
    in place.  Otherwise fall back to a single concatenation."""
    base = cz._base
    if (base is not None and cr._base is base and cq._base is base and base.is_contiguous() and base.dim() == 4
            and base.dtype == torch.float32):
        plane = base.shape[2] * base.shape[3]
        c = cz.shape[1]
        o0 = base.storage_offset()
        offs = [(t.storage_offset() - o0) for t in (cz, cr, cq)]
        if (offs[0] % plane == 0 and offs[1] == offs[0] + c * plane and offs[2] == offs[1] + c * plane
                and all(t.stride() == base.stride() for t in (cz, cr, cq))):
            return base, offs[0] // plane
    cat = torch.cat([cz, cr, cq], dim=1).float().contiguous()
    return cat, 0


class ConvGRU(nn.Module):
    def __init__(self, hidden_dim, input_dim, kernel_size=3):
        super().__init__()
        pad = kernel_size // 2
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=pad)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=pad)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=pad)
        self._pzr = ops.PackedConv()
        self._pzr_h, self._pzr_x = ops.PackedConv(), ops.PackedConv()
        self._pq = ops.PackedConv()
        self.fused_gates = __import__("os").environ.get("ANYSTEREO_FUSED_GATES", "1") != "0"  # training: gate math as two HIP stages (grad.GruGatesZR / GruGatesQ) instead of ~30 pointwise ops
        self.tag = "gru"  # timing label; BasicMultiUpdateBlock renames it gru04 / gru08 / gru16

    def forward(self, h, cz, cr, cq, *x_list, pre_zr=None):
        if _train(h, cz, cr, cq, self.convz.weight, *x_list):  # update.py:33-41
            x = torch.cat(x_list, dim=1)
            hx = torch.cat([h, x], dim=1)
            zr = _cs(self, "zr", hx, (self.convz.weight, self.convr.weight), (self.convz.bias, self.convr.bias))
            hid = h.shape[1]
            if zr.is_cuda and self.fused_gates:
                base, coff = _context_window(cz, cr, cq)
                hc = _f(h)
                # the context is the same tensor in every iteration: its gradient is accumulated by the gate stages' backward kernels
                # (grad.ContextAnchor) when it is the tensor cz / cr / cq are views of; the views stay inputs for the graph otherwise
                tok, holder = G.context_anchor(self, base) if base is cz._base else (base, None)
                z, rh = G.GruGatesZR.apply(zr, cz, cr, hc, tok, coff, holder)
                ql = _cs(self, "q", torch.cat([rh, x], dim=1), self.convq.weight, self.convq.bias)
                return G.GruGatesQ.apply(ql, cq, z, hc, tok, coff + 2 * hid, holder)
            z = torch.sigmoid(zr[:, :hid] + cz)
            r = torch.sigmoid(zr[:, hid:] + cr)
            q = torch.tanh(_cs(self, "q", torch.cat([r * h, x], dim=1), self.convq.weight, self.convq.bias) + cq)
            return (1 - z) * h + z * q
        hsrc = _twin(h) if isinstance(h, torch.Tensor) else h  # blocked twin written by the previous step's q conv
        h = _f(h)
        xs = [x if isinstance(x, ops.BS8) else _twin(_f(x)) for x in x_list]
        ctx, coff = _context_window(cz, cr, cq)
        hid = h.shape[1]
        links = _links()
        b, _, hh, ww = h.shape
        # r*h goes to the q conv as a blocked tensor only; the new hidden state gets a blocked twin for its conv consumers
        # (next step's gate conv, the disparity head) next to the fp32 tensor the epilogues / resamplers read
        rbs = ops.BS8.empty(b, hid, hh, ww, h.device) if links else None
        hbs = ops.BS8.empty(b, hid, hh, ww, h.device) if links else None
        pq = self._pq.get([self.convq.weight], [self.convq.bias])
        if pre_zr is None:
            pzr = self._pzr.get([self.convz.weight, self.convr.weight], [self.convz.bias, self.convr.bias])
            with scope(self.tag + "_zr_conv"):
                z, rh = ops.conv2d([hsrc if links else h] + xs, pzr, add=ctx, add_coff=coff, epilogue=L.EPI_GRU_ZR, h=h,
                                   out_bs=rbs, bs_only=links)
        else:
            pzx = self._pzr_x.get([self.convz.weight, self.convr.weight], [None, None], transform=lambda w: w[:, hid:])
            with scope(self.tag + "_zr_conv"):
                z, rh = ops.conv2d(xs, pzx, add=pre_zr, add_coff=0, epilogue=L.EPI_GRU_ZR, h=h, out_bs=rbs, bs_only=links)
        ops.mark_fine(self.tag + "_zr_end")
        with scope(self.tag + "_q_conv"):
            out = ops.conv2d([rbs if links else rh] + xs, pq, add=ctx, add_coff=coff + 2 * hid, epilogue=L.EPI_GRU_Q, h=h, z=z,
                             out_bs=hbs)
        if links:
            _attach_twin(out, hbs)
        return out

    def pre_zr(self, h, cz, cr, cq):
        """The part of convz‖convr that needs only the hidden state: conv([h], W[:, :hidden]) + bias + [cz‖cr].  The inference
        schedule (models/base.py) issues it at the top of an iteration, while the motion features are still being computed
        on the other stream; forward(..., pre_zr=) then finishes the gates with the x-part of the weights (one fp32 add of two
        partial sums instead of one K loop: same arithmetic up to summation order)."""
        h = _f(h)
        ctx, coff = _context_window(cz, cr, cq)
        hid = h.shape[1]
        pzh = self._pzr_h.get([self.convz.weight, self.convr.weight], [self.convz.bias, self.convr.bias],
                              transform=lambda w: w[:, :hid])
        with scope(self.tag + "_zr_pre"):
            return ops.conv2d([h], pzh, add=ctx, add_coff=coff)


class BasicMotionEncoder(nn.Module):
    def __init__(self, args, geo_channels=8):
        super().__init__()
        self.args = args
        # geo_channels: 8 (IGEV, coreContinuous_IGEV/update.py:77) or 0 (RAFT, corePrune_RAFT/update.py:77)
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) * (geo_channels + 1)
        self.convc1 = nn.Conv2d(cor_planes, 64, 1, padding=0)
        self.convc2 = nn.Conv2d(64, 64, 3, padding=1)
        self.convd1 = nn.Conv2d(1, 64, 7, padding=3)
        self.convd2 = nn.Conv2d(64, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 64, 128 - 1, 3, padding=1)
        self._pc1, self._pc2, self._pd2, self._pc = (ops.PackedConv() for _ in range(4))


    def forward(self, disp, corr):
        if _train(disp, corr, self.convc1.weight):  # update.py:84-92
            cor = _cs(self, "c1", corr, self.convc1.weight, self.convc1.bias, relu=True)
            cor = _cs(self, "c2", cor, self.convc2.weight, self.convc2.bias, relu=True)
            dsp = _cs(self, "d2", G.conv7x7_c1_relu(self, "d1", disp, self.convd1), self.convd2.weight, self.convd2.bias, relu=True)
            out = _cs(self, "c", torch.cat([cor, dsp], dim=1), self.conv.weight, self.conv.bias, relu=True)
            return torch.cat([out, disp], dim=1)
        disp, corr = _f(disp), _f(corr)
        cd, out = self.new_buffer(disp), self.new_output(disp)
        if self.dual_branches and ops.get_precision() == "split":
            # convc1 and convd1, then convc2 and convd2 (same shape, independent inputs) as ONE launch into the two halves of cd
            bs = isinstance(cd, ops.BS8)
            b, _, h, w = disp.shape
            with scope("enc_convc1"):
                if bs:
                    cor = ops.BS8.empty(b, 64, h, w, corr.device)
                    ops.conv2d([corr], self._pc1.get([self.convc1.weight], [self.convc1.bias]), act=L.ACT_RELU, out_bs=cor, bs_only=True)
                else:
                    cor = ops.conv2d([corr], self._pc1.get([self.convc1.weight], [self.convc1.bias]), act=L.ACT_RELU)
            with scope("enc_convd1"):
                d1 = ops.conv7x7_c1_relu(disp, self.convd1.weight, self.convd1.bias,
                                         out=ops.BS8.empty(b, 64, h, w, disp.device) if bs else None, copy_out=out, copy_coff=127)
            with scope("enc_convc2"):
                second = {"src": d1, "pack": self._pd2.get([self.convd2.weight], [self.convd2.bias]), "out_coff": 64, "out_bs_coff": 64}
                pc2 = self._pc2.get([self.convc2.weight], [self.convc2.bias])
                if bs:
                    ops.conv2d([cor], pc2, act=L.ACT_RELU, out_bs=cd, out_bs_coff=0, bs_only=True, dual=second)
                else:
                    ops.conv2d([cor], pc2, act=L.ACT_RELU, out=cd, out_coff=0, dual=second)
            return self.merge(cd, disp, out)
        self.corr_branch(corr, cd)
        self.disp_branch(disp, cd, out)
        return self.merge(cd, disp, out)

    dual_branches = __import__("os").environ.get("ANYSTEREO_DUAL_BRANCHES", "1") != "0"
    fused_lookup = __import__("os").environ.get("ANYSTEREO_FUSED_LOOKUP", "1") != "0"
    def fused_lookup_ok(self, lookup_fn) -> bool:
        return (self.fused_lookup and self.dual_branches and _links() and not torch.is_grad_enabled()
                and getattr(lookup_fn, "fused_convc1_ok", None) is not None and lookup_fn.fused_convc1_ok()
                and self.convc1.out_channels == 64 and self.convc1.kernel_size == (1, 1))

    # Opt-in fusions of the front of an iteration (ANYSTEREO_FUSED_FRONT), both bit-identical to the staged launches and both
    # measured NOT faster on cfg 2, so the default stays staged ("0"):
    #   "lite": the head's finish rides in the lookup + convc1 launch (the lookup blocks derive the new disparity from the 36
    #           tap planes themselves), the 7x7 conv stays its own launch — 48.4-48.6 vs 49.4 pairs/s staged on the same box;
    #   "full": the 7x7 conv's blocks in the same grid as well — 28.3 us against 27.9 us for the three staged launches (the
    #           lookup blocks alone fill the chip for 16 us, so the 7x7 blocks run as a second wave rather than beside them).
    fused_front = {"0": False, "1": "full", "lite": "lite", "full": "full"}[__import__("os").environ.get("ANYSTEREO_FUSED_FRONT", "0")]

    def forward_front(self, taps, head, disp_old, lookup_fn):
        """The head's finish (disp_old + delta), the fused lookup + convc1 and the 7x7 conv of the disparity branch as ONE launch,
        then convc2 || convd2 and the merge conv: -> (motion features, new disparity).  Same arithmetic as
        head.finish -> forward_fused_lookup (the new disparity is bit-identical)."""
        disp_old = _f(disp_old)
        cd, out = self.new_buffer(disp_old), self.new_output(disp_old)
        if not hasattr(self, "_plc1"):
            self._plc1 = ops.LookupConvPack()
        full = self.fused_front == "full"
        disp, cor, d1 = lookup_fn.loop_front(taps, head.conv2.bias, disp_old, self._plc1.get(self.convc1.weight, self.convc1.bias),
                                             self.convd1.weight if full else None, self.convd1.bias if full else None, out, 127)
        if not full:
            d1 = ops.BS8.empty(disp.shape[0], 64, disp.shape[2], disp.shape[3], disp.device)
            with scope("enc_convd1"):
                ops.conv7x7_c1_relu(disp, self.convd1.weight, self.convd1.bias, out=d1, copy_out=out, copy_coff=127)
        with scope("enc_convc2"):
            second = {"src": d1, "pack": self._pd2.get([self.convd2.weight], [self.convd2.bias]), "out_coff": 64, "out_bs_coff": 64}
            ops.conv2d([cor], self._pc2.get([self.convc2.weight], [self.convc2.bias]), act=L.ACT_RELU, out_bs=cd, out_bs_coff=0,
                       bs_only=True, dual=second)
        return self.merge(cd, disp, out), disp

    def forward_fused_lookup(self, disp, lookup_fn):
        """forward(disp, lookup_fn(disp)) with the lookup fused into convc1 (one kernel, blocked split-fp16 result): the
        [B,162,h,w] correlation features are never written (update.py:84-92 with geometry.py:34-60 inlined)."""
        disp = _f(disp)
        cd, out = self.new_buffer(disp), self.new_output(disp)
        b, _, h, w = disp.shape
        if not hasattr(self, "_plc1"):
            self._plc1 = ops.LookupConvPack()
        cor = ops.BS8.empty(b, 64, h, w, disp.device)
        d1 = ops.BS8.empty(b, 64, h, w, disp.device)
        # (Tried and removed: the 7x7 conv on a branch stream beside the fused lookup — both depend on `disp` only — forked
        # from the loop's side stream: capturing that nested fork into the forward's hipGraph crashed the process.)
        lookup_fn.lookup_convc1(disp, self._plc1.get(self.convc1.weight, self.convc1.bias), out_bs=cor)
        ops.mark_fine("enc.lookup_end")
        with scope("enc_convd1"):
            ops.conv7x7_c1_relu(disp, self.convd1.weight, self.convd1.bias, out=d1, copy_out=out, copy_coff=127)
        ops.mark_fine("enc.conv7x7_end")
        with scope("enc_convc2"):
            second = {"src": d1, "pack": self._pd2.get([self.convd2.weight], [self.convd2.bias]), "out_coff": 64, "out_bs_coff": 64}
            ops.conv2d([cor], self._pc2.get([self.convc2.weight], [self.convc2.bias]), act=L.ACT_RELU, out_bs=cd, out_bs_coff=0,
                       bs_only=True, dual=second)
        ops.mark_fine("enc.dual_end")
        return self.merge(cd, disp, out)

    # The three pieces of forward(), exposed so the inference schedule (models/base.py::_iterate_pipelined) can run
    # the two independent branches on different streams.  cd [B,128,h,w]: channels [0,64) = correlation branch,
    # [64,128) = disparity branch (the reference's torch.cat, update.py:90, never materialised separately).
    def new_buffer(self, disp):
        b, _, h, w = disp.shape
        if _links():  # the two branches meet in a blocked tensor that only the last conv reads
            return ops.BS8.empty(b, 128, h, w, disp.device)
        return torch.empty((b, 128, h, w), device=disp.device, dtype=torch.float32)

    def new_output(self, disp):
        b, _, h, w = disp.shape
        if _links():  # the motion features only feed gru04's two convolutions
            return ops.BS8.empty(b, 128, h, w, disp.device)
        return torch.empty((b, 128, h, w), device=disp.device, dtype=torch.float32)

    def corr_branch(self, corr, cd):
        bs = isinstance(cd, ops.BS8)
        with scope("enc_convc1"):
            if bs:
                b, _, h, w = corr.shape
                cor = ops.BS8.empty(b, 64, h, w, corr.device)
                ops.conv2d([_f(corr)], self._pc1.get([self.convc1.weight], [self.convc1.bias]), act=L.ACT_RELU, out_bs=cor, bs_only=True)
            else:
                cor = ops.conv2d([_f(corr)], self._pc1.get([self.convc1.weight], [self.convc1.bias]), act=L.ACT_RELU)
        with scope("enc_convc2"):
            if bs:
                ops.conv2d([cor], self._pc2.get([self.convc2.weight], [self.convc2.bias]), act=L.ACT_RELU, out_bs=cd, out_bs_coff=0, bs_only=True)
            else:
                ops.conv2d([cor], self._pc2.get([self.convc2.weight], [self.convc2.bias]), act=L.ACT_RELU, out=cd, out_coff=0)

    def disp_branch(self, disp, cd, out=None):
        """convd1 -> convd2 into cd[:, 64:]; with `out` the disparity itself is also placed in out[:, 127] (update.py:91)."""
        with scope("enc_convd1"):
            d1 = ops.conv7x7_c1_relu(_f(disp), self.convd1.weight, self.convd1.bias,
                                     copy_out=out, copy_coff=127)
        with scope("enc_convd2"):
            if isinstance(cd, ops.BS8):
                ops.conv2d([d1], self._pd2.get([self.convd2.weight], [self.convd2.bias]), act=L.ACT_RELU, out_bs=cd, out_bs_coff=64, bs_only=True)
            else:
                ops.conv2d([d1], self._pd2.get([self.convd2.weight], [self.convd2.bias]), act=L.ACT_RELU, out=cd, out_coff=64)

    def merge(self, cd, disp, out=None):
        """relu(conv(cd)) into channels [0,127) and disp in channel 127; `out` given = disp_branch already placed disp."""
        have_disp = out is not None
        if out is None:
            out = self.new_output(disp)
        with scope("enc_conv"):
            if isinstance(out, ops.BS8):  # channels [0,127) here, channel 127 (disp) by the 7x7 kernel's pass-through
                ops.conv2d([cd], self._pc.get([self.conv.weight], [self.conv.bias]), act=L.ACT_RELU, out_bs=out, out_bs_coff=0, bs_only=True)
            else:
                ops.conv2d([cd], self._pc.get([self.conv.weight], [self.conv.bias]), act=L.ACT_RELU, out=out, out_coff=0)
        if not have_disp:
            if isinstance(out, ops.BS8):
                raise RuntimeError("BasicMotionEncoder.merge: a blocked output needs disp_branch(..., out=) to place the disparity")
            out[:, 127:128].copy_(_f(disp))
        return out


def _hip_train_input(x, what):
    """Training inputs of the resamplers: fp32 CUDA tensors go to the HIP forward / backward kernels as they are; fp16 / bf16
    CUDA tensors (the reference's autocast training, train_continuous_IGEV.py:206,288) are cast to fp32 for the kernel and the
    result is cast back; anything else is an error — the hot path has no library / CPU fallback."""
    if not x.is_cuda:
        raise RuntimeError(f"anystereo {what}: expected a CUDA tensor (the hot path has no CPU fallback), got {x.device}")
    if x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise RuntimeError(f"anystereo {what}: dtype {x.dtype} (float32, or float16 / bfloat16 under autocast)")
    return x if x.dtype == torch.float32 else x.float()


def pool2x(x):
    if _train(x):  # update.py:94-95 (3x3 average pool, stride 2, zero padding 1 counted in the divisor)
        with scope("pool2x"):
            return G.Pool2x.apply(_hip_train_input(x, "pool2x")).to(x.dtype)
    with scope("pool2x"):
        return ops.pool2x_bs(_f(x)) if _links() else ops.pool2x(_f(x))  # the pooled / resized maps only feed GRU convs


def interp(x, dest):
    if _train(x):  # update.py:100-102 (bilinear resize to dest's size, align_corners=True)
        with scope("interp"):
            return G.InterpBilinear.apply(_hip_train_input(x, "interp"), dest.shape[2], dest.shape[3]).to(x.dtype)
    with scope("interp"):
        if _links():
            return ops.interp_bs(_f(x), dest.shape[2], dest.shape[3])
        return ops.interp(_f(x), dest.shape[2], dest.shape[3])


class BasicMultiUpdateBlock(nn.Module):
    def __init__(self, args, hidden_dims=[], geo_channels=8):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(args, geo_channels)
        encoder_output_dim = 128
        self.gru04 = ConvGRU(hidden_dims[2], encoder_output_dim + hidden_dims[1] * (args.n_gru_layers > 1))
        self.gru08 = ConvGRU(hidden_dims[1], hidden_dims[0] * (args.n_gru_layers == 3) + hidden_dims[2])
        self.gru16 = ConvGRU(hidden_dims[0], hidden_dims[1])
        self.disp_head = DispHead(hidden_dims[2], hidden_dim=256, output_dim=1)
        self.gru04.tag, self.gru08.tag, self.gru16.tag = "gru04", "gru08", "gru16"

    # The motion encoder (5 convs at 1/4 res) does not depend on the 1/16 and 1/8 GRUs, and those two are far
    # too small to fill 256 CUs: run the encoder on a second HIP stream, fork/join with events (captured as
    # parallel branches under hipGraph).  Same arithmetic, same results; set `parallel_encoder=False` to serialise.
    parallel_encoder = True

    def _side_stream(self, device, index: int = 0):
        streams = self.__dict__.setdefault("_streams", {})
        if (device, index) not in streams:
            streams[(device, index)] = torch.cuda.Stream(device=device)
        return streams[(device, index)]

    def forward(self, net, inp, corr=None, disp=None, iter04=True, iter08=True, iter16=True, update=True):
        motion_features = None
        side = None
        if (self.parallel_encoder and iter04 and (iter08 or iter16) and corr is not None and corr.is_cuda
                and not torch.is_grad_enabled()):
            main = torch.cuda.current_stream(corr.device)
            side = self._side_stream(corr.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                motion_features = self.encoder(disp, corr)
        if iter16:
            net[2] = self.gru16(net[2], *(inp[2]), pool2x(net[1]))
        if iter08:
            if self.args.n_gru_layers > 2:
                net[1] = self.gru08(net[1], *(inp[1]), pool2x(net[0]), interp(net[2], net[1]))
            else:
                net[1] = self.gru08(net[1], *(inp[1]), pool2x(net[0]))
        if iter04:
            if side is not None:
                main.wait_stream(side)
                motion_features.record_stream(main)
            else:
                motion_features = self.encoder(disp, corr)
            if self.args.n_gru_layers > 1:
                net[0] = self.gru04(net[0], *(inp[0]), motion_features, interp(net[1], net[0]))
            else:
                net[0] = self.gru04(net[0], *(inp[0]), motion_features)
        if not update:
            return net
        delta_disp = self.disp_head(net[0])
        return net, delta_disp
