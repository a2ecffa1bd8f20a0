"""The bottleneck stages of the ResNet backbone (res2 .. res5: 52 convolutions of R50) as ONE autograd node on hand-written kernels,
in the style of swin_core.py / functions/encoder_core.py.

Reference: detectron2 0.6 `BottleneckBlock.forward` (un-vendored; selected by configs/mask2former/coco/instance-segmentation/
Base-COCO-InstanceSegmentation.yaml:2-15):  conv1 1x1 -> FrozenBN -> ReLU, conv2 3x3 (stride here) -> FrozenBN ->
ReLU, conv3 1x1 -> FrozenBN, + shortcut (identity | 1x1 conv -> FrozenBN), ReLU.

Forward: one pd_igemm_bf16 launch per convolution with the frozen-BN affine, the residual add and the ReLU in its epilogue
(include/pd_igemm.h) — 52 launches where the module path issues a library convolution + an epilogue kernel each.
Backward: one launch per convolution too.  With g = dL/d(pre-activation) of a layer, its input gradient is g . (scale (.) W) — the
scale is folded into the TRANSPOSED filter copy the input-gradient GEMM needs anyway (one grouped transpose launch per step) — and
the launch's epilogue adds the gradient arriving over the other branch (identity shortcut: dense; strided shortcut convolution:
its COMPACT [B, H/2, W/2, C] input gradient, added at even pixels only), adds the gradient that enters from outside (res2 .. res4
feed the pixel decoder) and applies the ReLU mask of the layer below, so what it writes IS the g of the layer below.  Filter
gradients: dW = scale (.) (g^T x), all 52 in the grouped transpose-read launch of csrc/conv_bf16.hip with the scale in its reduce.
All activations and gradients live in ONE persistent arena per input shape: every pointer is known when the plan is built, the
launch lists are built once, and a step costs the host ONE call per direction (pd_igemm_bf16_seq issues the launches from C++).
"""
import ctypes

import torch
from torch.autograd import Function

from ... import lib as _lib
from ...compat.layers import FrozenBatchNorm2d
from ...functions import conv_bf16, igemm
from ...functions.fused import PinnedRing

# one Plan (= one persistent arena holding every activation, gradient, transposed filter and filter gradient of the body: ~3.5 GB at
# 2 x 1024^2) per (network, input shape, grad mode) — bounded, least recently used first: the reference's mappers crop / resize to
# varying sizes and inference shapes differ per batch
from ... import cmdbuf as _cmdbuf
_PLANS = _cmdbuf.LRU(int(__import__("os").environ.get("PD_R50_PLAN_CAP", "4")))
ENABLED = bool(int(__import__("os").environ.get("PD_R50_FUSED", "1")))
# Data-parallel runs: `PUBLISH(parameter)` is the gradient reducer's per-parameter "gradient complete" entry (engine/ddp.py sets it while a
# reducer with > 1 participant is alive).  The node then walks its backward STAGE BY STAGE (res5 -> res2): a stage's input-gradient launches,
# that stage's grouped filter-gradient launch, `.grad` of its filters set and published — so the buckets holding res5 / res4 / res3 leave for the
# wire while the lower stages still compute, instead of all ~94 MB after the last launch of the body.  None: one sequence, one
# grouped filter-gradient launch, gradients returned to autograd (the single-GPU path: 4 C calls instead of 10).
PUBLISH = None
PER_STAGE = bool(int(__import__("os").environ.get("PD_R50_STAGE_HANDOVER", "1")))


def _align(n, a=128):
    return (n + a - 1) // a * a


class _Conv:
    __slots__ = ("mod", "ci", "co", "k", "stride", "pad", "scale", "bias", "wt_off", "dw_off")


class Plan:
    """arena layout + launch lists of the bottleneck body for one input shape"""

    def __init__(self, blocks, x0, out_names, stage_of_block, grad):
        from .resnet import BottleneckBlock
        dev = x0.device
        B, C0, H, W = x0.shape
        self.dev, self.shape, self.grad = dev, (B, C0, H, W), grad
        self.generation = 0
        self.blocks = blocks
        total = 0

        def take(n):
            nonlocal total
            o = total
            total += _align(n)
            return o

        # ---- geometry + arena offsets (bf16 elements)
        info = []
        h, w, cin = H, W, C0
        for bi, blk in enumerate(blocks):
            assert isinstance(blk, BottleneckBlock)
            s = blk.conv2.stride[0] if blk.conv1.stride[0] == 1 else None
            mid, cout = blk.conv1.out_channels, blk.conv3.out_channels
            ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
            d = dict(h=h, w=w, ho=ho, wo=wo, cin=cin, mid=mid, cout=cout, s=s, sc=blk.shortcut is not None)
            d["a"], d["b"], d["out"] = take(B * h * w * mid), take(B * ho * wo * mid), take(B * ho * wo * cout)
            d["scv"] = take(B * ho * wo * cout) if d["sc"] else None
            if grad:
                d["g_a"], d["g_b"], d["g_out"] = take(B * h * w * mid), take(B * ho * wo * mid), take(B * ho * wo * cout)
                d["g_sc"] = take(B * ho * wo * cin) if d["sc"] else None
            info.append(d)
            h, w, cin = ho, wo, cout
        self.info = info
        self.gx0_off = take(B * H * W * C0) if grad else None
        # ---- convolutions: frozen-BN affine, transposed (scaled) filters, filter-gradient targets
        convs = []
        for blk in blocks:
            for name in ("conv1", "conv2", "conv3", "shortcut"):
                m = getattr(blk, name)
                if m is None:
                    convs.append(None)
                    continue
                c = _Conv()
                c.mod, c.ci, c.co, c.k, c.stride, c.pad = m, m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], m.padding[0]
                sc, bi_ = m.norm.scale_bias()
                c.scale, c.bias = sc.float().contiguous(), bi_.float().contiguous()
                n = m.weight.numel()
                c.wt_off = take(n) if grad else None
                c.dw_off = take(n) if grad else None
                convs.append(c)
        self.convs = convs                                  # 4 per block: conv1, conv2, conv3, shortcut | None
        self.arena = torch.empty(total, dtype=torch.bfloat16, device=dev)
        self.base = self.arena.data_ptr()
        self.ones = torch.ones(max(d["cout"] for d in info), dtype=torch.float32, device=dev)
        self.stage_of_block = list(stage_of_block)
        self.out_blocks = {}                                # stage name -> index of its last block
        for bi, st in enumerate(stage_of_block):
            self.out_blocks[st] = bi
        self.out_names = [n for n in out_names if n in self.out_blocks]
        self.norm_version = self._norm_version()
        self.w_ptrs = self._weight_ptrs()
        self._build_forward()
        if grad:
            self._build_backward()

    # ------------------------------------------------------------------ helpers
    def _norm_version(self):
        v = 0
        for c in self.convs:
            if c is not None:
                v += c.mod.norm.weight._version + c.mod.norm.running_var._version
        return v

    def _weight_ptrs(self):
        return tuple(c.mod.weight.data_ptr() for c in self.convs if c is not None)

    def valid(self):
        return self._weight_ptrs() == self.w_ptrs and self._norm_version() == self.norm_version

    def ptr(self, off):
        return self.base + 2 * off

    def owns(self, ptr):
        return self.base <= ptr < self.base + 2 * self.arena.numel()

    def view(self, off, b, h, w, c):
        """NCHW-shaped view with channels_last strides of the arena's [b][h][w][c] block at `off`"""
        return self.arena.as_strided((b, c, h, w), (h * w * c, 1, w * c, c), off)

    def weights(self):
        return [c.mod.weight for c in self.convs if c is not None]

    # ------------------------------------------------------------------ launch lists
    def _build_forward(self):
        B = self.shape[0]
        lst, self.fwd_x0_entries = [], []
        x_off = None                                        # None: the node's input x0 (patched per call)
        for bi, d in enumerate(self.info):
            c1, c2, c3, cs = self.convs[4 * bi:4 * bi + 4]

            def ent(c, src, hs, ws, ho, wo, out, res=None, relu=True):
                e = igemm.PdIgemm()
                e.src = self.ptr(src) if src is not None else None
                e.w, e.scale, e.bias = c.mod.weight.data_ptr(), c.scale.data_ptr(), c.bias.data_ptr()
                e.res = self.ptr(res) if res is not None else None
                e.out = self.ptr(out)
                e.batch, e.hs, e.ws, e.cs, e.ho, e.wo, e.n = B, hs, ws, c.ci, ho, wo, c.co
                e.k, e.stride, e.pad, e.dgrad = c.k, c.stride, c.pad, 0
                e.act = igemm.ACT_RELU if relu else igemm.ACT_NONE
                if src is None:
                    self.fwd_x0_entries.append(len(lst))
                lst.append(e)

            ent(c1, x_off, d["h"], d["w"], d["h"], d["w"], d["a"])
            ent(c2, d["a"], d["h"], d["w"], d["ho"], d["wo"], d["b"])
            if cs is not None:
                ent(cs, x_off, d["h"], d["w"], d["ho"], d["wo"], d["scv"], relu=False)
            ent(c3, d["b"], d["ho"], d["wo"], d["ho"], d["wo"], d["out"], res=d["scv"] if cs is not None else x_off)
            if cs is None and x_off is None:
                raise RuntimeError("the first bottleneck block must have a shortcut convolution")
            x_off = d["out"]
        self.fwd = (igemm.PdIgemm * len(lst))(*lst)
        L = _lib.load()
        for i in self.fwd_x0_entries:
            self.fwd[i].src = self.base                     # any valid address: the workspace query looks at geometry only
        need = int(L.pd_igemm_bf16_seq_workspace_bytes(self.fwd, len(lst)))
        if need < 0:
            raise _lib.PdHipError(L.pd_last_error().decode())
        self.ws_need = need

    def _build_backward(self):
        B = self.shape[0]
        lst, ext_entries = [], {}
        n = len(self.info)
        self.tr = []                                        # (conv) of every transposed filter
        wg = []                                             # filter-gradient descriptors (ctypes) + which need x0
        self.wg_x0 = []
        self.stage_ranges = []                              # per stage, last first: [name, first launch, end launch, first filter, end filter]
        for bi in reversed(range(n)):
            d = self.info[bi]
            st_name = self.stage_of_block[bi]
            if not self.stage_ranges or self.stage_ranges[-1][0] != st_name:
                self.stage_ranges.append([st_name, len(lst), len(lst), len(wg), len(wg)])
            c1, c2, c3, cs = self.convs[4 * bi:4 * bi + 4]
            prev = self.info[bi - 1] if bi > 0 else None

            def ent(c, src, hs, ws, ho, wo, out, *, dgrad_k, stride=1, gate=None, res=None, res_mode=igemm.RES_DENSE):
                """input gradient of convolution c: source = g at its output [B, hs, ws, co], result grid (ho, wo) with ci channels"""
                e = igemm.PdIgemm()
                e.src, e.w = self.ptr(src), self.ptr(c.wt_off)
                e.res = self.ptr(res) if res is not None else None
                e.gate = self.ptr(gate) if gate is not None else None
                e.gate_mode = igemm.GATE_RELU if gate is not None else igemm.GATE_NONE
                e.res_mode = res_mode
                e.out = self.ptr(out)
                e.batch, e.hs, e.ws, e.cs, e.ho, e.wo, e.n = B, hs, ws, c.co, ho, wo, c.ci
                e.k, e.stride, e.pad, e.dgrad = dgrad_k, stride, dgrad_k // 2, 1 if dgrad_k > 1 else 0
                lst.append(e)
                return len(lst) - 1

            ent(c3, d["g_out"], d["ho"], d["wo"], d["ho"], d["wo"], d["g_b"], dgrad_k=1, gate=d["b"])
            ent(c2, d["g_b"], d["ho"], d["wo"], d["h"], d["w"], d["g_a"], dgrad_k=3, stride=d["s"], gate=d["a"])
            # the gradient of the block's input: conv1's input gradient + the shortcut's; written as the g of the block below
            if cs is not None:
                ent(cs, d["g_out"], d["ho"], d["wo"], d["ho"], d["wo"], d["g_sc"], dgrad_k=1)          # compact when the shortcut strides
                res, mode = d["g_sc"], (igemm.RES_UP2 if d["s"] == 2 else igemm.RES_DENSE)
            else:
                res, mode = d["g_out"], igemm.RES_DENSE
            dst = prev["g_out"] if prev is not None else self.gx0_off
            i = ent(c1, d["g_a"], d["h"], d["w"], d["h"], d["w"], dst, dgrad_k=1, gate=prev["out"] if prev is not None else None, res=res, res_mode=mode)
            if prev is not None:
                for name, last in self.out_blocks.items():
                    if last == bi - 1 and name in self.out_names:
                        ext_entries[name] = i               # the external gradient of that stage output enters here (res2)
            for c, dz, x, hi, wi, ho, wo in ((c1, d["g_a"], prev["out"] if prev else None, d["h"], d["w"], d["h"], d["w"]),
                                             (c2, d["g_b"], d["a"], d["h"], d["w"], d["ho"], d["wo"]),
                                             (c3, d["g_out"], d["b"], d["ho"], d["wo"], d["ho"], d["wo"]),
                                             (cs, d["g_out"], prev["out"] if prev else None, d["h"], d["w"], d["ho"], d["wo"])):
                if c is None:
                    continue
                self.tr.append(c)
                w = conv_bf16._Desc()
                w.dz, w.x, w.dw, w.db = self.ptr(dz), (self.ptr(x) if x is not None else None), self.ptr(c.dw_off), None
                w.batch, w.hi, w.wi, w.ci, w.ho, w.wo, w.co, w.k, w.stride, w.pad = B, hi, wi, c.ci, ho, wo, c.co, c.k, c.stride, c.pad
                w.scale = c.scale.data_ptr()
                if x is None:
                    self.wg_x0.append(len(wg))
                wg.append((c, w))
            self.stage_ranges[-1][2], self.stage_ranges[-1][4] = len(lst), len(wg)
        self.bwd = (igemm.PdIgemm * len(lst))(*lst)
        self.ext_entries = ext_entries
        self.last_name = next((nm for nm, last in self.out_blocks.items() if last == n - 1), None)
        L = _lib.load()
        need = int(L.pd_igemm_bf16_seq_workspace_bytes(self.bwd, len(lst)))
        if need < 0:
            raise _lib.PdHipError(L.pd_last_error().decode())
        self.ws_need = max(self.ws_need, need)
        # transposed, scaled filters: one grouped launch per step
        self.tr_descs = (igemm.PdFilterTranspose * len(self.tr))()
        for t, c in zip(self.tr_descs, self.tr):
            t.src, t.dst, t.scale, t.co, t.taps, t.ci = c.mod.weight.data_ptr(), self.ptr(c.wt_off), c.scale.data_ptr(), c.co, c.k * c.k, c.ci
        tb = int(L.pd_filter_transpose_table_bytes(len(self.tr)))
        self.tr_ring = PinnedRing(tb, torch.uint8, pin=True)
        self.tr_dev = torch.empty(tb, dtype=torch.uint8, device=self.dev)
        self.wg_all = wg
        self._wg_cache = {}

    # ------------------------------------------------------------------ execution
    def _ws(self):
        return igemm.workspace(self.dev, self.ws_need) if self.ws_need else None

    def run_forward(self, x0):
        p = x0.data_ptr()
        for i in self.fwd_x0_entries:
            self.fwd[i].src = p
        blk0_res = None
        ws = self._ws()
        L = _lib.load()
        _lib.check(L.pd_igemm_bf16_seq(self.fwd, len(self.fwd), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                       _lib.current_stream()))
        self.generation += 1
        del blk0_res
        B = self.shape[0]
        outs = []
        for name in self.out_names:
            d = self.info[self.out_blocks[name]]
            outs.append(self.view(d["out"], B, d["ho"], d["wo"], d["cout"]))
        return outs

    def _wgrad_descs(self, mask, lo=0, hi=None):
        """ctypes descriptor array of the filters [lo, hi) of wg_all whose gradient is wanted (cached per requires-grad pattern and range)"""
        hi = len(self.wg_all) if hi is None else hi
        key = (mask, lo, hi)
        hit = self._wg_cache.get(key)
        if hit is None:
            sel = [(i, w) for i, ((c, w), m) in enumerate(zip(self.wg_all, mask)) if m and lo <= i < hi]
            arr = (conv_bf16._Desc * len(sel))()
            for j, (_, w) in enumerate(sel):
                ctypes.memmove(ctypes.addressof(arr[j]), ctypes.addressof(w), ctypes.sizeof(conv_bf16._Desc))
            x0_idx = [j for j, (i, _) in enumerate(sel) if i in self.wg_x0]
            hit = self._wg_cache[key] = (arr, x0_idx)
        return hit

    def run_backward(self, x0, gouts, need_x0, need_w):
        L = _lib.load()
        st = _lib.current_stream()
        B = self.shape[0]
        # transposed + scaled filters of this step's weights
        host = self.tr_ring.acquire()
        _lib.check(L.pd_filter_transpose_grouped(self.tr_descs, len(self.tr), host.data_ptr(), self.tr_dev.data_ptr(), st))
        self.tr_ring.release()
        keep = []
        gmap = dict(zip(self.out_names, gouts))

        def nhwc(g, d):
            g = g if g.dtype == torch.bfloat16 else g.to(torch.bfloat16)
            g = g if g.is_contiguous(memory_format=torch.channels_last) else g.contiguous(memory_format=torch.channels_last)
            keep.append(g)
            return g

        # g of the last block = incoming gradient * ReLU mask of its output
        dl = self.info[-1]
        g_last = gmap.get(self.last_name)
        gl_view = self.view(dl["g_out"], B, dl["ho"], dl["wo"], dl["cout"])
        if g_last is None:
            gl_view.zero_()
        else:
            g_last = nhwc(g_last, dl)
            _lib.check(L.pd_affine_act_bwd_bf16(g_last.data_ptr(), self.ptr(dl["out"]), self.ones.data_ptr(), self.ptr(dl["g_out"]), None,
                                                g_last.numel(), dl["cout"], 1, st))
        for name, i in self.ext_entries.items():
            g = gmap.get(name)
            self.bwd[i].res2 = nhwc(g, None).data_ptr() if g is not None else None
        count = len(self.bwd) if need_x0 else len(self.bwd) - (2 if self.convs[3] is not None else 1)
        ws = self._ws()
        dws = [None] * len(need_w)
        order = [c for c, _ in self.wg_all]
        pos = {id(c): j for j, c in enumerate(c for c in self.convs if c is not None)}
        mask = tuple(bool(need_w[pos[id(c)]]) for c in order)
        publish = PUBLISH if (PER_STAGE and any(need_w)) else None
        esz = ctypes.sizeof(igemm.PdIgemm)

        def launches(a, b):
            b = min(b, count)
            if b > a:
                sub = (igemm.PdIgemm * (b - a)).from_address(ctypes.addressof(self.bwd) + a * esz)
                _lib.check(L.pd_igemm_bf16_seq(sub, b - a, ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, st))

        def filter_gradients(lo, hi):
            """one grouped launch (per tile shape) over the convolutions [lo, hi) of wg_all that want a gradient -> their positions"""
            arr, x0_idx = self._wgrad_descs(mask, lo, hi)
            n = len(arr)
            if n == 0:
                return []
            p0 = x0.data_ptr()
            for j in x0_idx:
                arr[j].x = p0
            need = int(L.pd_conv_bf16_wgrad_grouped_workspace_floats(arr, n))
            wsf = conv_bf16._WS.get(str(self.dev))
            if wsf is None or wsf.numel() < need:
                wsf = conv_bf16._WS[str(self.dev)] = torch.empty(max(need, 1 << 24), dtype=torch.float32, device=self.dev)
            D = conv_bf16._Deferred
            tbytes = int(L.pd_conv_bf16_wgrad_grouped_table_bytes(D.MAXP))
            if D.ring is None:
                D.ring = PinnedRing(tbytes, torch.uint8, pin=True)
            if D.table_dev is None or D.table_dev.device != self.dev:
                D.table_dev = torch.empty(tbytes, dtype=torch.uint8, device=self.dev)
            hostt = D.ring.acquire()
            rc = L.pd_conv_bf16_wgrad_grouped(arr, n, hostt.data_ptr(), D.table_dev.data_ptr(), wsf.data_ptr(), wsf.numel(), st)
            D.ring.release()
            _lib.check(rc)
            done = []
            for i in range(lo, hi):
                c = order[i]
                j = pos[id(c)]
                if need_w[j]:
                    wt = c.mod.weight
                    dws[j] = self.arena.as_strided(wt.shape, wt.stride(), c.dw_off)
                    done.append(j)
            return done

        if publish is None:
            launches(0, len(self.bwd))
            if any(need_w):
                filter_gradients(0, len(order))
        else:
            # stage by stage, last stage first: its gradients are complete (and on their way) while the stages below still compute
            weights = self.weights()
            for _, a, b, lo, hi in self.stage_ranges:
                launches(a, b)
                for j in filter_gradients(lo, hi):
                    p = weights[j]
                    if p.grad is None and publish(p, dws[j]):    # handed over here: autograd gets no gradient for it (no second accumulation)
                        dws[j] = None
        gx0 = self.view(self.gx0_off, *[self.shape[i] for i in (0, 2, 3, 1)]) if need_x0 else None
        del keep
        return gx0, dws


class R50Body(Function):
    @staticmethod
    def forward(ctx, x0, plan, *weights):
        outs = plan.run_forward(x0)
        ctx.plan, ctx.gen = plan, plan.generation
        ctx.save_for_backward(x0)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        plan = ctx.plan
        if plan.generation != ctx.gen:
            raise RuntimeError("the fused ResNet body ran another forward before this backward: its activation arena was overwritten "
                               "(set PD_R50_FUSED=0 for graphs that keep several forward passes alive)")
        (x0,) = ctx.saved_tensors
        _cmdbuf.unalias_grads(plan.weights(), plan.owns)             # a `.grad` that survived the last step still points at the arena's dW
        gx0, dws = plan.run_backward(x0, gouts, ctx.needs_input_grad[0], ctx.needs_input_grad[2:])
        return (gx0, None, *dws)


def supported(blocks, x0):
    from .resnet import BottleneckBlock
    if not (ENABLED and x0.is_cuda and x0.dtype == torch.bfloat16 and x0.dim() == 4 and x0.is_contiguous(memory_format=torch.channels_last)
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return False
    h, w, cin = x0.shape[2], x0.shape[3], x0.shape[1]
    for bi, blk in enumerate(blocks):
        if not isinstance(blk, BottleneckBlock):
            return False
        if bi == 0 and blk.shortcut is None:
            return False
        for name in ("conv1", "conv2", "conv3", "shortcut"):
            m = getattr(blk, name)
            if m is None:
                continue
            if not (isinstance(m.norm, FrozenBatchNorm2d) and m.weight.dtype == torch.bfloat16 and m.groups == 1 and m.dilation[0] == 1
                    and m.bias is None and m.in_channels % 64 == 0 and m.out_channels % 64 == 0
                    and (m.weight.is_contiguous(memory_format=torch.channels_last) or m.kernel_size[0] == 1)):
                return False
        if blk.conv1.stride[0] != 1 or blk.conv1.kernel_size[0] != 1 or blk.conv2.kernel_size[0] != 3 or blk.conv2.padding[0] != 1 \
                or blk.conv3.kernel_size[0] != 1 or blk.conv3.stride[0] != 1:
            return False
        s = blk.conv2.stride[0]
        if s not in (1, 2) or (blk.shortcut is not None and (blk.shortcut.stride[0] != s or blk.shortcut.kernel_size[0] != 1)):
            return False
        if blk.shortcut is None and (s != 1 or blk.conv1.in_channels != blk.conv3.out_channels):
            return False
        if s == 2 and (h % 2 or w % 2):
            return False
        h, w, cin = (h - 1) // s + 1, (w - 1) // s + 1, blk.conv3.out_channels
    return True


def run_body(resnet, x0, blocks, stage_of_block):
    """-> {stage name: map} for the stages in resnet._out_features (the stem output is handled by the caller)"""
    grad = torch.is_grad_enabled()
    key = (id(resnet), tuple(x0.shape), str(x0.device), grad)
    plan = _PLANS.get(key)
    if plan is None or not plan.valid():
        plan = Plan(blocks, x0, [n for n in resnet._out_features if n != "stem"], stage_of_block, grad)
        _PLANS.put(key, plan)
    outs = R50Body.apply(x0, plan, *plan.weights()) if grad else plan.run_forward(x0)
    return dict(zip(plan.out_names, outs))
