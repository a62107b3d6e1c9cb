"""Host-side description of the GEMM-shaped layers: operand packing and the forward / data-gradient /
weight-gradient calls into libzs_amd.so.  PyTorch is used for device memory only."""
import ctypes
import os

import torch

from . import _lib as L


def colsum_ok(T):
    """zs_gemm_conv can add the per-sample column sums of its output (ZsGemmConv.colsum) when whole samples fit its tiles."""
    return T > 0 and 128 % T == 0


def rup(x, m):
    return (x + m - 1) // m * m


DTYPES = {'fp32': (L.ZS_F32, torch.float32), 'bf16': (L.ZS_BF16, torch.bfloat16)}
SLACK = 512   # elements of zeroed slack behind every activation buffer (K-padding reads may run past a row)


class Act(object):
    """A channels-last activation: B*T rows of `ld` elements inside tensor `t` (1-D storage with slack),
    starting at element offset `off`.  C valid channels; columns [C, cols) are zeros."""
    __slots__ = ('t', 'B', 'T', 'C', 'ld', 'off', 'cols')

    def __init__(self, t, B, T, C, ld, off=0, cols=None):
        self.t, self.B, self.T, self.C, self.ld, self.off = t, B, T, C, ld, off
        self.cols = cols if cols is not None else ld

    @property
    def rows(self):
        return self.B * self.T

    def ptr(self, col=0):
        return self.t.data_ptr() + (self.off + col) * self.t.element_size()

    def sub(self, col, C, cols=None):
        return Act(self.t, self.B, self.T, C, self.ld, self.off + col, cols if cols is not None else C)

    def view2d(self):
        return self.t[self.off:self.off + self.rows * self.ld].view(self.rows, self.ld) if self.off == 0 else \
            self.t[self.off - (self.off % self.ld):][:self.rows * self.ld].view(self.rows, self.ld)

    def valid(self):
        """[B, T, C] torch view of the valid part (for API outputs / tests)."""
        base = self.t[self.off:self.off + (self.rows - 1) * self.ld + self.C]
        return torch.as_strided(base, (self.B, self.T, self.C), (self.T * self.ld, self.ld, 1))


_SIDE = {}
N_SIDE = max(1, int(os.environ.get('ZS_SIDE_STREAMS', '4')))


def side_stream(device):
    """Extra HIP streams per device for work that is off the critical path (weight gradients): N_SIDE streams used round
    robin, so that the many small weight-gradient GEMMs (conv bank, T'=16 layers) run side by side instead of in a queue."""
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = {'streams': [torch.cuda.Stream(device) for _ in range(N_SIDE)], 'used': False, 'next': 0, 'ws': [None] * N_SIDE}
    return _SIDE[key]


def side_events(device):
    """One event per side stream, recorded now (everything enqueued on the side streams so far)."""
    sd = side_stream(device)
    evs = []
    for st in sd['streams']:
        ev = torch.cuda.Event()
        ev.record(st)
        evs.append(ev)
    return evs


def fork_side(device):
    """Make every side stream wait for what is enqueued on the current stream so far; returns the side streams.  Used to run
    independent small launches of the main chain (the encoder's 7-conv bank) side by side; close with join_side()."""
    sd = side_stream(device)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    for st in sd['streams']:
        st.wait_event(ev)
    sd['used'] = True
    return sd['streams']


def join_side(device):
    """Make the current stream wait for everything enqueued on the side streams."""
    sd = side_stream(device)
    if sd['used']:
        cur = torch.cuda.current_stream(device)
        for ev in side_events(device):
            cur.wait_event(ev)
        sd['used'] = False
    sd['next'] = 0          # the same launch -> stream assignment every step (hipGraph capture replays exactly this)


_OPT = {}


def opt_stream(device):
    """A third stream: the decoder's clip + Adam + re-pack run here while the encoder's backward is still on the main one."""
    key = (device.type, device.index)
    if key not in _OPT:
        _OPT[key] = torch.cuda.Stream(device)
    return _OPT[key]


_STATUS = {}


def device_status(device):
    """The sticky status word of a device (ZsGruFwd.status in zs_amd.h), shared by every Ctx on it."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _STATUS:
        _STATUS[key] = torch.zeros(4, dtype=torch.int32, device=device)
    return _STATUS[key]


def check_status(device):
    """Read the device's sticky status word (synchronises: call where the host syncs anyway) and raise if a persistent GRU
    pass timed out since the last check -- the outputs / gradients of that step were invalid.  Clears the word."""
    st = device_status(device)
    v = int(st[0].item())
    if v:
        st.zero_()
        raise L.ZsError('persistent GRU %s timed out waiting for its workgroup group: the results since the last check are '
                        'invalid (status 0x%x; is another process holding CUs of this GPU?)' %
                        (' and '.join(n for b, n in ((1, 'forward'), (2, 'BPTT')) if v & b), v))


_COPY = {}


def copy_stream(device):
    """The stream the in-graph H2D copy of the next batch runs on (trainer.HostFedStep)."""
    key = (device.type, device.index)
    if key not in _COPY:
        _COPY[key] = torch.cuda.Stream(device)
    return _COPY[key]


class Ctx(object):
    """Per-model execution context: device, compute dtype, cached buffers, split-K workspace."""

    def __init__(self, device, dtype='fp32'):
        if dtype not in DTYPES:
            raise ValueError('dtype must be fp32 or bf16')
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise L.ZsError('zs_amd runs on an MI355X only (device %s requested); there is no CPU path' % device)
        L.lib()
        self.dtype_name = dtype
        self.dt, self.tdt = DTYPES[dtype]
        self.es = 4 if dtype == 'fp32' else 2
        self.kc = 128 // self.es
        self._bufs = {}
        self._ws = None
        # True: a layer's weight gradients always go to the same side stream (stage 2: several passes of a step accumulate into
        # one gradient buffer, and stream order makes that sum deterministic)
        self.pin_wgrad_streams = False
        # sticky status word of the persistent kernels (ZsGruFwd.status): OR-ed into on a bounded-spin timeout, never cleared by
        # the library; read at the host's own sync points by check_status()
        self.status = device_status(self.device)
        # weight gradients are only needed by the optimizer: run them on a second stream under the latency-bound
        # phases of the backward chain (GRU steps, small norms)
        self.overlap_wgrad = os.environ.get('ZS_OVERLAP_WGRAD', '1') == '1'

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def raw(self, name, n, dtype, zero=True):
        key = (name, n, dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros(n + SLACK, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def act(self, name, B, T, C, ld=None, dtype=None):
        ld = ld if ld is not None else rup(C, 32)
        dtype = dtype if dtype is not None else self.tdt
        return Act(self.raw(name, B * T * ld, dtype), B, T, C, ld)

    def f32(self, name, n):
        return self.raw(name, n, torch.float32)

    def workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            if self._ws is not None:
                torch.cuda.synchronize(self.device)      # the side stream may still be using the old buffer (rare: growth)
            self._ws = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=self.device)
        return self._ws

    def release(self):
        self._bufs.clear()
        self._ws = None

    def check_status(self):
        check_status(self.device)


class ConvLayer(object):
    """Conv1d [Cout,Cin,k] or Linear [Cout,Cin] (k=1) as packed MFMA operands.
    split2: output channels packed so that the epilogue can pixel-shuffle (see zs_amd.h)."""

    def __init__(self, ctx, weight, bias, gweight, gbias, stride=1, split2=False, pad_mode=L.ZS_PAD_REFLECT, name='', padded=True,
                 strides=None):
        self.ctx, self.name = ctx, name
        self.w, self.b, self.gw, self.gb = weight, bias, gweight, gbias
        self.Cout, self.Cin = weight.shape[0], weight.shape[1]
        self.k = weight.shape[2] if weight.dim() == 3 else 1
        self.stride, self.split2, self.pad_mode = stride, split2, pad_mode
        self.pad_l = self.k // 2 if padded else 0              # pad_layer(): (k//2, k//2 - 1 | k//2); plain nn.Conv1d: none
        self.pad_r = (self.k - 1 - self.k // 2) if padded else 0
        self.so, self.si, self.sj = (self.Cin * self.k, self.k, 1) if weight.dim() == 3 else (self.Cin, 1, 0)
        if strides is not None:                           # a column block of a wider parameter (element strides of weight / gweight)
            self.so, self.si, self.sj = strides
        kc = ctx.kc                                       # elements per 128-byte K chunk
        self.cin_pad, self.cout_pad = rup(self.Cin, kc), rup(self.Cout, kc)
        self.ldw, self.n_pad = self.k * self.cin_pad, rup(self.Cout, 128)
        self.ldw_d, self.n_pad_d = self.k * self.cout_pad, rup(self.Cin, 128)
        dev, tdt = ctx.device, ctx.tdt
        self.wf = torch.zeros(self.n_pad * self.ldw + SLACK, dtype=tdt, device=dev)
        self.wd = torch.zeros(self.n_pad_d * self.ldw_d + SLACK, dtype=tdt, device=dev)
        self.bias_p = torch.zeros(self.Cout, dtype=torch.float32, device=dev) if (split2 and bias is not None) else None
        self.sid = next_sid()          # the side stream this layer's weight gradients always use
        # stride-2 data gradient by output parity: position u of the padded domain only receives the taps j = u (mod 2), so the
        # transposed convolution is two stride-1 correlations (even rows: taps 0, 2, 4; odd rows: taps 1, 3) over half the rows
        # each -- half the MFMA work of feeding zero rows for the other taps (see dgrad)
        self.parity = (stride == 2 and self.k > 1)

    def pack(self):
        c = self.ctx
        common = dict(dtype=c.dt, W=L.ptr(self.w), so=self.so, si=self.si, sj=self.sj, Cout=self.Cout, Cin=self.Cin,
                      taps=self.k, co_split2=int(self.split2))
        L.call('zs_pack_weight', 'ZsPackWeight', c.stream, transpose=0, inner_pad=self.cin_pad, dst=L.ptr(self.wf),
               ldw=self.ldw, n_rows=self.n_pad, n_cols=self.ldw, **common)
        if self.parity:
            # stride 2: taps in parity order [j = 0, 2, 4, .. | j = 1, 3, ..] (see dgrad): two strided views of the parameter
            ne, no = (self.k + 1) // 2, self.k // 2
            ev = dict(common, taps=ne, sj=2 * self.sj)
            od = dict(common, taps=no, sj=2 * self.sj, W=L.ptr(self.w, self.sj))
            L.call('zs_pack_weight', 'ZsPackWeight', c.stream, transpose=1, inner_pad=self.cout_pad, dst=L.ptr(self.wd),
                   ldw=self.ldw_d, n_rows=self.n_pad_d, n_cols=ne * self.cout_pad, **ev)
            L.call('zs_pack_weight', 'ZsPackWeight', c.stream, transpose=1, inner_pad=self.cout_pad, dst=L.ptr(self.wd),
                   ldw=self.ldw_d, n_rows=self.n_pad_d, n_cols=no * self.cout_pad, col_offset=ne * self.cout_pad, **od)
        else:
            L.call('zs_pack_weight', 'ZsPackWeight', c.stream, transpose=1, inner_pad=self.cout_pad, dst=L.ptr(self.wd),
                   ldw=self.ldw_d, n_rows=self.n_pad_d, n_cols=self.ldw_d, **common)
        if self.bias_p is not None:
            h = self.Cout // 2
            L.vec_copy(self.bias_p[:h], self.b, c.stream, src_stride=2)
            L.vec_copy(self.bias_p[h:], self.b[1:], c.stream, src_stride=2)

    def bias_ptr(self):
        if self.b is None:
            return None
        return L.ptr(self.bias_p if self.bias_p is not None else self.b)

    def t_out(self, T_in):
        return (T_in + self.pad_l + self.pad_r - self.k) // self.stride + 1

    def fwd(self, A, out=None, act=L.ZS_ACT_NONE, slope=0.0, out_f32=False, out_cols=None, store_mode=L.ZS_STORE_ROWS,
            out2=None, vec2=None, idx=None, store_mode2=L.ZS_STORE_ROWS, out2_cols=None, pre_vec=None, bias=True, lengths=None):
        """A: Act [B,T_in,Cin] -> out Act [B,T_out,Cout] (or pixel-shuffled).  Returns T_out.
        lengths: int32 device tensor [B], valid input rows per sample (ragged batch, see ZsGemmConv.lengths)."""
        c = self.ctx
        T_out = self.t_out(A.T)
        kw = dict(dtype=c.dt, A=A.ptr(), lda=A.ld, a_batch_stride=A.T * A.ld, B=A.B, T_in=A.T, T_out=T_out, taps=self.k,
                  stride=self.stride, pad_left=self.pad_l, pad_mode=self.pad_mode, gather=0, cin_pad=self.cin_pad,
                  W=L.ptr(self.wf), ldw=self.ldw, N=self.Cout, n_pad=self.n_pad, bias=(self.bias_ptr() if bias else None), act=act,
                  slope=slope, groups=1)
        if pre_vec is not None:
            kw.update(pre_vec=L.ptr(pre_vec), pre_vec_ld=pre_vec.shape[1], vec_idx=L.ptr(idx))
        if lengths is not None and self.k > 1:             # (a 1-tap layer is row-wise: rows past a length stay unread garbage)
            kw.update(lengths=L.ptr(lengths), pad_right=self.pad_r)
        if out is not None:
            kw.update(out=out.ptr(), ldc=out.ld, out_f32=int(out_f32), store_mode=store_mode,
                      out_cols=out_cols if out_cols is not None else (min(out.cols, rup(self.Cout, 32)) if store_mode == L.ZS_STORE_ROWS else self.Cout // 2))
        if out2 is not None:
            kw.update(out2=out2.ptr(), ldc2=out2.ld, store_mode2=store_mode2,
                      out2_cols=out2_cols if out2_cols is not None else (min(out2.cols, rup(self.Cout, 32)) if store_mode2 == L.ZS_STORE_ROWS else self.Cout // 2))
            if vec2 is not None:
                kw.update(vec2=L.ptr(vec2), vec2_ld=vec2.shape[1], vec_idx=L.ptr(idx))
        L.call('zs_gemm_conv', 'ZsGemmConv', c.stream, **kw)
        return T_out

    def dgrad(self, dY, T_x, out, dact_src=None, slope=0.0, add_src=None, out_f32=False, add_f32=False, colsum=None, out_cols=None,
              n_cols=None, n_off=0, colsum_post=False):
        """dY: Act [B,T_y,Cout] (ld >= cout_pad).  out: Act with B*(T_x+pad_l+pad_r) rows (padded domain; equals the
        input gradient when k == 1).  Optional epilogue: *lrelu'(dact_src), +add_src (both only meaningful for k == 1).
        colsum = (fp32 pointer, ld, col0): per-sample column sums of the raw gradient, columns >= col0 (see colsum_ok).
        n_cols (, n_off): only the gradient of the n_cols input channels from n_off on is wanted (the rest is neither computed nor
        stored; column 0 of `out` is channel n_off).  colsum_post: the column sums are taken of the stored value (after mask / add)."""
        c = self.ctx
        Tp = T_x + self.pad_l + self.pad_r
        if self.parity:
            return self._dgrad_parity(dY, Tp, out, out_f32, out_cols)
        N, n_pad = self.Cin, self.n_pad_d
        if n_cols is not None and (n_cols < self.Cin or n_off):
            N = n_cols
            n_pad = rup(n_cols, 256) if n_off + rup(n_cols, 256) <= self.n_pad_d else rup(n_cols, 128)
            if n_off + n_pad > self.n_pad_d:
                raise ValueError('dgrad: channel block [%d, %d) does not fit the packed weight (%d rows)' % (n_off, n_off + n_cols, self.n_pad_d))
        kw = dict(dtype=c.dt, A=dY.ptr(), lda=dY.ld, a_batch_stride=dY.T * dY.ld, B=dY.B, T_in=dY.T, T_out=Tp, taps=self.k,
                  stride=self.stride, pad_left=0, pad_mode=L.ZS_PAD_ZERO, gather=1, cin_pad=self.cout_pad,
                  W=L.ptr(self.wd, n_off * self.ldw_d), ldw=self.ldw_d, N=N, n_pad=n_pad, act=L.ZS_ACT_NONE, slope=slope,
                  out=out.ptr(), ldc=out.ld, out_cols=(out_cols if out_cols is not None else min(out.cols, rup(N, 32))),
                  store_mode=L.ZS_STORE_ROWS, groups=1, out_f32=int(out_f32))
        if dact_src is not None:
            kw.update(dact_src=dact_src.ptr(), dact_ld=dact_src.ld)
        if add_src is not None:
            kw.update(add_src=add_src.ptr(), add_ld=add_src.ld, add_f32=int(add_f32))
        if colsum is not None:
            kw.update(colsum=colsum[0], colsum_ld=colsum[1], colsum_col0=colsum[2], colsum_post=int(colsum_post))
        L.call('zs_gemm_conv', 'ZsGemmConv', c.stream, **kw)
        return Tp

    def _dgrad_parity(self, dY, Tp, out, out_f32, out_cols):
        """Stride-2 data gradient as two stride-1 launches (rows u = 2v and u = 2v + 1 of the padded domain).  `out` holds
        Te = Tp rounded up to even rows per sample; the extra row of an odd domain receives no valid tap (zeros)."""
        c = self.ctx
        Te = Tp + (Tp & 1)
        if out.T != Te:
            raise ValueError('stride-2 dgrad: the output needs %d rows per sample (padded domain %d rounded up to even), got %d' % (Te, Tp, out.T))
        ne = (self.k + 1) // 2
        for par, taps in ((0, ne), (1, self.k // 2)):
            kw = dict(dtype=c.dt, A=dY.ptr(), lda=dY.ld, a_batch_stride=dY.T * dY.ld, B=dY.B, T_in=dY.T, T_out=Te // 2, taps=taps,
                      stride=1, pad_left=0, pad_mode=L.ZS_PAD_ZERO, gather=1, cin_pad=self.cout_pad,
                      W=L.ptr(self.wd, par * ne * self.cout_pad), ldw=self.ldw_d, N=self.Cin, n_pad=self.n_pad_d, act=L.ZS_ACT_NONE,
                      out=out.ptr(par * out.ld), ldc=2 * out.ld, out_cols=(out_cols if out_cols is not None else min(out.cols, rup(self.Cin, 32))),
                      store_mode=L.ZS_STORE_ROWS, groups=1, out_f32=int(out_f32))
            L.call('zs_gemm_conv', 'ZsGemmConv', c.stream, **kw)
        return Tp

    def wgrad(self, dY, X, accumulate=False, bias=True):
        """gw (+)= dY^T * gather(X);  gb (+)= column sums of dY (same kernel, no atomics; bias=False leaves gb alone)."""
        c = self.ctx
        es = c.es
        y_cols = min(dY.cols, rup(self.Cout, 16 // es))
        x_cols = min(X.cols, rup(self.Cin, 16 // es))
        kw = dict(dtype=c.dt, dY=dY.ptr(), ldy=dY.ld, y_cols=y_cols, X=X.ptr(), ldx=X.ld, x_batch_stride=X.T * X.ld,
                  x_cols=x_cols, B=X.B, T_in=X.T, T_out=dY.T, taps=self.k, stride=self.stride, pad_left=self.pad_l,
                  pad_mode=self.pad_mode, Cout=self.Cout, Cin=self.Cin, dW=L.ptr(self.gw), so=self.so, si=self.si, sj=self.sj,
                  db=(L.ptr(self.gb) if bias else None), co_split2=int(self.split2), accumulate=int(accumulate), splits=0)
        wgrad_call(c, kw, self.sid)


class Conv2dLayer(object):
    """nn.Conv2d [Cout, Cin, k, k] (stride 1 or 2, padding k // 2 by reflect or zero pad on both axes) on channels-last rows
    [B, H*W, C], as ONE implicit GEMM per pass with 2-D taps in the kernels' row pointers (ZsGemmConv.w_in): no im2col buffer.
    The weight's dims (2, 3) run along W and H of the rows layout (for the stage-2 critic: frequency and time), i.e. tap
    j = kw*k + kh is the parameter's own flat order -- forward operand and weight gradient address the parameter directly.
    Stride-2 data gradient by output parity in BOTH axes: four stride-1 transposed correlations (k = 5: 3x3, 3x2, 2x3, 2x2 taps) over
    the padded input domain, one class buffer each, joined and un-padded by zs_conv2d_unpad.
    Measured against the stage-2 critic's gathered path (im2col along H + Conv1d over W; tools/conv2d_bench.py, B = 128, bf16, us,
    gathered path incl. its gather / fold kernels in brackets): forward layer 2..5: 536 (476), 414 (361), 420 (375), 200 (224);
    data gradient 1092 (1027), 519 (583), 483 (568), 255 (400); weight gradient layer 2: 1052 (817).  The pointer arithmetic of
    a tap change (every 128-byte chunk at C = 64) and the half-empty 128-column tiles at C = 64 cost more than the im2col traffic
    saves, and in the D step the 2-D data gradients of layers 3-5 gave 31.3 against 30.9 ms: zs_amd.patch keeps the gathered
    path; this class is the tested entry to the 2-D mode of the kernels (tests/test_gpu_kernels.py::test_conv2d_layer_vs_torch)."""

    def __init__(self, ctx, weight, bias, gweight, gbias, stride=2, pad_mode=L.ZS_PAD_REFLECT, name='', dgrad_only=False):
        self.ctx, self.name = ctx, name
        self.w, self.b, self.gw, self.gb = weight, bias, gweight, gbias
        self.Cout, self.Cin, self.k = weight.shape[0], weight.shape[1], weight.shape[2]
        if weight.dim() != 4 or weight.shape[3] != self.k or stride not in (1, 2):
            raise ValueError('Conv2dLayer: square kernels, stride 1 or 2')
        self.stride, self.pad_mode, self.pad = stride, pad_mode, self.k // 2
        self.taps = self.k * self.k
        kc = ctx.kc
        self.cin_pad, self.cout_pad = rup(self.Cin, kc), rup(self.Cout, kc)
        self.ldw, self.n_pad, self.n_pad_d = self.taps * self.cin_pad, rup(self.Cout, 128), rup(self.Cin, 128)
        dev, tdt = ctx.device, ctx.tdt
        self.wf = None if dgrad_only else torch.zeros(self.n_pad * self.ldw + SLACK, dtype=tdt, device=dev)      # forward operand
        # data-gradient operands: per parity class (ph, pw) rows = input channels, K = (kw', kh') x cout_pad
        self.nt = [(self.k + 1) // 2, self.k // 2] if stride == 2 else [self.k]
        self.wd = {}
        for ph in range(len(self.nt)):
            for pw in range(len(self.nt)):
                ld = self.nt[ph] * self.nt[pw] * self.cout_pad
                self.wd[(ph, pw)] = (torch.zeros(self.n_pad_d * ld + SLACK, dtype=tdt, device=dev), ld)
        self.sid = next_sid()

    def out_hw(self, H, W):
        return (H + 2 * self.pad - self.k) // self.stride + 1, (W + 2 * self.pad - self.k) // self.stride + 1

    def pack(self):
        c, k = self.ctx, self.k
        common = dict(dtype=c.dt, so=self.Cin * self.taps, si=self.taps, Cout=self.Cout, Cin=self.Cin, co_split2=0)
        if self.wf is not None:
            L.call('zs_pack_weight', 'ZsPackWeight', c.stream, transpose=0, inner_pad=self.cin_pad, dst=L.ptr(self.wf), ldw=self.ldw,
                   n_rows=self.n_pad, n_cols=self.ldw, W=L.ptr(self.w), sj=1, taps=self.taps, **common)
        step = self.stride
        for (ph, pw), (buf, ld) in self.wd.items():
            nh, nw = self.nt[ph], self.nt[pw]
            for kwp in range(nw):                                 # real taps kw = pw + step*kw', kh = ph + step*kh'
                L.call('zs_pack_weight', 'ZsPackWeight', c.stream, transpose=1, inner_pad=self.cout_pad, dst=L.ptr(buf), ldw=ld,
                       n_rows=self.n_pad_d, n_cols=nh * self.cout_pad, col_offset=kwp * nh * self.cout_pad,
                       W=L.ptr(self.w, (pw + step * kwp) * k + ph), sj=step, taps=nh, **common)

    def fwd(self, X, H, W, out, act=L.ZS_ACT_NONE, slope=0.0, bias=True, out_f32=False):
        """X: Act [B, H*W, Cin] -> out: Act [B, Ho*Wo, Cout].  Returns (Ho, Wo)."""
        c = self.ctx
        Ho, Wo = self.out_hw(H, W)
        if self.wf is None:
            raise ValueError('Conv2dLayer.fwd: built with dgrad_only')
        if X.T != H * W or out.T != Ho * Wo or X.B != out.B:
            raise ValueError('Conv2dLayer.fwd: rows %d x %d for an image of %d x %d, output %d x %d' % (X.B, X.T, H, W, out.B, out.T))
        L.call('zs_gemm_conv', 'ZsGemmConv', c.stream, dtype=c.dt, A=X.ptr(), lda=X.ld, a_batch_stride=X.T * X.ld, B=X.B, T_in=H * W,
               T_out=Ho * Wo, taps=self.taps, stride=self.stride, pad_left=self.pad, pad_mode=self.pad_mode, gather=0, cin_pad=self.cin_pad,
               W=L.ptr(self.wf), ldw=self.ldw, N=self.Cout, n_pad=self.n_pad, bias=(L.ptr(self.b) if (bias and self.b is not None) else None),
               act=act, slope=slope, groups=1, out=out.ptr(), ldc=out.ld, out_f32=int(out_f32), store_mode=L.ZS_STORE_ROWS,
               out_cols=min(out.cols, rup(self.Cout, 32)), w_in=W, w_out=Wo, taps_h=self.k)
        return Ho, Wo

    def wgrad(self, dY, X, H, W, accumulate=False, bias=True):
        """gw (+)= dY^T * taps(X);  gb (+)= column sums of dY.  dY: Act [B, Ho*Wo, Cout]; X: Act [B, H*W, Cin]."""
        c, es = self.ctx, self.ctx.es
        Ho, Wo = self.out_hw(H, W)
        kw = dict(dtype=c.dt, dY=dY.ptr(), ldy=dY.ld, y_cols=min(dY.cols, rup(self.Cout, 16 // es)), X=X.ptr(), ldx=X.ld,
                  x_batch_stride=X.T * X.ld, x_cols=min(X.cols, rup(self.Cin, 16 // es)), B=X.B, T_in=H * W, T_out=Ho * Wo, taps=self.taps,
                  stride=self.stride, pad_left=self.pad, pad_mode=self.pad_mode, Cout=self.Cout, Cin=self.Cin, dW=L.ptr(self.gw),
                  so=self.Cin * self.taps, si=self.taps, sj=1, db=(L.ptr(self.gb) if bias else None), co_split2=0,
                  accumulate=int(accumulate), splits=0, w_in=W, w_out=Wo, taps_h=self.k)
        wgrad_call(c, kw, self.sid)

    def padded_hw(self, H, W):
        """Extent of the padded input domain the forward pass reads."""
        Ho, Wo = self.out_hw(H, W)
        return (Ho - 1) * self.stride + self.k, (Wo - 1) * self.stride + self.k

    def dgrad(self, dY, H, W, out, name, add=None, fill_cols=None):
        """dY: Act [B, Ho*Wo, Cout] -> out: Act [B, H*W, Cin] = gradient w.r.t. the layer input (+ add), stride 2."""
        c = self.ctx
        if self.stride != 2:
            raise ValueError('Conv2dLayer.dgrad: stride 2 only')
        Ho, Wo = self.out_hw(H, W)
        Hp, Wp = self.padded_hw(H, W)
        g = {}
        for (ph, pw), (buf, ld) in self.wd.items():
            Hc, Wc = (Hp - ph + 1) // 2, (Wp - pw + 1) // 2
            gq = c.act('%s_q%d%d' % (name, ph, pw), dY.B, Hc * Wc, self.Cin)
            L.call('zs_gemm_conv', 'ZsGemmConv', c.stream, dtype=c.dt, A=dY.ptr(), lda=dY.ld, a_batch_stride=dY.T * dY.ld, B=dY.B,
                   T_in=Ho * Wo, T_out=Hc * Wc, taps=self.nt[ph] * self.nt[pw], stride=1, pad_left=0, pad_mode=L.ZS_PAD_ZERO, gather=1,
                   cin_pad=self.cout_pad, W=L.ptr(buf), ldw=ld, N=self.Cin, n_pad=self.n_pad_d, act=L.ZS_ACT_NONE, out=gq.ptr(), ldc=gq.ld,
                   out_cols=min(gq.cols, rup(self.Cin, 32)), store_mode=L.ZS_STORE_ROWS, groups=1, out_f32=0, w_in=Wo, w_out=Wc,
                   taps_h=self.nt[ph])
            g[(ph, pw)] = gq
        g00 = g[(0, 0)]
        L.call('zs_conv2d_unpad', 'ZsConv2dUnpad', c.stream, dtype=c.dt, g00=g00.ptr(), g01=g[(0, 1)].ptr(), g10=g[(1, 0)].ptr(),
               g11=g[(1, 1)].ptr(), ldg=g00.ld, B=dY.B, H=H, W=W, C=rup(self.Cin, 8), Hp=Hp, Wp=Wp, pad=self.pad, pad_mode=self.pad_mode,
               add=(add.ptr() if add is not None else None), ldadd=(add.ld if add is not None else 0), out=out.ptr(), ldo=out.ld,
               fill_cols=(fill_cols if fill_cols is not None else out.ld))


_SID = [0]


def next_sid():
    _SID[0] += 1
    return _SID[0]


def side_launch(ctx, sid, need):
    """Where a weight-gradient launch goes: (stream handle, workspace of >= need bytes).  With ctx.overlap_wgrad a side stream made
    to wait for everything enqueued so far on the main stream (dY and X are complete), asynchronous from then on; the launches of one
    side stream serialise, so they share that stream's workspace.  sid: None = round robin over the side streams; with
    ctx.pin_wgrad_streams an int pins the call to side stream sid % N (every weight gradient of one parameter on ONE stream:
    accumulating launches are ordered behind each other)."""
    if not ctx.overlap_wgrad:
        return ctx.stream, ctx.workspace(need)
    sd = side_stream(ctx.device)
    if sid is not None and ctx.pin_wgrad_streams:
        i = sid % len(sd['streams'])
    else:
        i = sd['next']                               # plain round robin (measured: 1 stream 13.7 ms/step, 3: 13.5, 4: 13.0,
        sd['next'] = (i + 1) % len(sd['streams'])    # 8: 13.05; big GEMMs pinned to one stream: 13.55)
    ws = sd['ws'][i]
    if ws is None or ws.numel() < need:
        # growth (first steps only): every side stream gets a workspace of the new size, so that no later launch --
        # in particular none inside a hipGraph capture -- has to grow one
        if any(w is not None for w in sd['ws']):
            torch.cuda.synchronize(ctx.device)      # the streams may still be using the old buffers
        size = int(need * 1.25) + 1024
        sd['ws'] = [torch.empty(size, dtype=torch.uint8, device=ctx.device) for _ in sd['streams']]
        ws = sd['ws'][i]
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(ctx.device))
    sd['streams'][i].wait_event(ev)
    sd['used'] = True
    return sd['streams'][i].cuda_stream, ws


def wgrad_call(ctx, kw, sid=None):
    """zs_gemm_wgrad through side_launch()."""
    S = L.STRUCTS['ZsGemmWgrad']
    s = S()
    for k, v in kw.items():
        if v is not None:
            setattr(s, k, v)
    need = L.lib().zs_gemm_wgrad_workspace_bytes(ctypes.byref(s))
    stream, ws = side_launch(ctx, sid, need)
    s.workspace = ws.data_ptr()
    s.workspace_bytes = ws.numel()
    L.check(L.lib().zs_gemm_wgrad(ctypes.byref(s), ctypes.c_void_p(stream)), 'zs_gemm_wgrad')


class GruLayer(object):
    """Bidirectional nn.GRU (single layer).  Parameters are the 8 nn.GRU tensors (and their grads)."""

    def __init__(self, ctx, P, G, prefix, name=''):
        self.ctx, self.name = ctx, name
        sfx = ['', '_reverse']
        self.w_ih = [P[prefix + 'weight_ih_l0' + s] for s in sfx]
        self.w_hh = [P[prefix + 'weight_hh_l0' + s] for s in sfx]
        self.b_ih = [P[prefix + 'bias_ih_l0' + s] for s in sfx]
        self.b_hh = [P[prefix + 'bias_hh_l0' + s] for s in sfx]
        self.gw_ih = [G[prefix + 'weight_ih_l0' + s] for s in sfx]
        self.gw_hh = [G[prefix + 'weight_hh_l0' + s] for s in sfx]
        self.gb_ih = [G[prefix + 'bias_ih_l0' + s] for s in sfx]
        self.gb_hh = [G[prefix + 'bias_hh_l0' + s] for s in sfx]
        self.H = self.w_hh[0].shape[1]
        self.Cin = self.w_ih[0].shape[1]
        self.sids = [next_sid() for _ in range(4)]
        H, Cin = self.H, self.Cin
        if H % 8:
            raise ValueError('GRU hidden size must be a multiple of 8 (got %d)' % H)
        dev, tdt = ctx.device, ctx.tdt
        self.G6 = 6 * H
        self.g6_pad = rup(6 * H, ctx.kc)
        # input projection, both directions stacked on the output axis: [6H][Cin]
        kc = ctx.kc
        self.cin_pad = rup(Cin, kc)
        self.ih_ldw, self.ih_npad = self.cin_pad, rup(6 * H, 128)
        self.wih_f = torch.zeros(self.ih_npad * self.ih_ldw + SLACK, dtype=tdt, device=dev)
        self.ih_ldw_d, self.ih_npad_d = self.g6_pad, rup(Cin, 128)
        self.wih_d = torch.zeros(self.ih_npad_d * self.ih_ldw_d + SLACK, dtype=tdt, device=dev)
        self.bih = torch.zeros(6 * H, dtype=torch.float32, device=dev)
        # recurrent: forward [2][n_pad(3H)][ldw(H)], transposed [2][n_pad(H)][ldw(3H)]
        self.hh_ldw, self.hh_npad = rup(H, kc), rup(3 * H, 128)
        self.whh_f = torch.zeros(2 * self.hh_npad * self.hh_ldw + SLACK, dtype=tdt, device=dev)
        self.hh_ldw_t, self.hh_npad_t = rup(3 * H, kc), rup(H, 128)
        self.whh_t = torch.zeros(2 * self.hh_npad_t * self.hh_ldw_t + SLACK, dtype=tdt, device=dev)
        self.bhh = torch.zeros(2 * 3 * H, dtype=torch.float32, device=dev)
        # fast recurrent path (zs_amd.h: zs_gru_fwd): rows of W_hh gate-interleaved per 32 hidden units
        self.fast = (H % 32 == 0)
        self.perm = None
        if self.fast:
            n = torch.arange(3 * H)
            hc, rem = n // 96, n % 96
            self.perm = ((rem // 32) * H + hc * 32 + (rem % 32)).to(torch.int32).to(dev)

    def pack(self):
        c, H, Cin = self.ctx, self.H, self.Cin
        for d in range(2):
            com = dict(dtype=c.dt, so=Cin, si=1, sj=0, Cout=3 * H, Cin=Cin, taps=1, co_split2=0, W=L.ptr(self.w_ih[d]))
            L.call('zs_pack_weight', 'ZsPackWeight', c.stream, transpose=0, inner_pad=self.cin_pad, dst=L.ptr(self.wih_f),
                   ldw=self.ih_ldw, n_rows=3 * H, n_cols=self.ih_ldw, row_offset=3 * H * d, **com)
            L.call('zs_pack_weight', 'ZsPackWeight', c.stream, transpose=1, inner_pad=3 * H, dst=L.ptr(self.wih_d),
                   ldw=self.ih_ldw_d, n_rows=self.ih_npad_d, n_cols=3 * H, col_offset=3 * H * d, **com)
            com = dict(dtype=c.dt, so=H, si=1, sj=0, Cout=3 * H, Cin=H, taps=1, co_split2=0, W=L.ptr(self.w_hh[d]))
            L.call('zs_pack_weight', 'ZsPackWeight', c.stream, transpose=0, inner_pad=rup(H, c.kc),
                   dst=L.ptr(self.whh_f, d * self.hh_npad * self.hh_ldw), ldw=self.hh_ldw,
                   n_rows=(3 * H if self.fast else self.hh_npad), n_cols=self.hh_ldw, row_perm=L.ptr(self.perm), **com)
            L.call('zs_pack_weight', 'ZsPackWeight', c.stream, transpose=1, inner_pad=rup(3 * H, c.kc),
                   dst=L.ptr(self.whh_t, d * self.hh_npad_t * self.hh_ldw_t), ldw=self.hh_ldw_t, n_rows=self.hh_npad_t,
                   n_cols=self.hh_ldw_t, **com)
            L.vec_copy(self.bih[3 * H * d:3 * H * (d + 1)], self.b_ih[d], c.stream)
            L.vec_copy(self.bhh[3 * H * d:3 * H * (d + 1)], self.b_hh[d], c.stream)

    def _work(self, B):
        n = L.lib().zs_gru_work_bytes(B, self.H)
        return self.ctx.f32('gru_work_%s' % self.name, (n + 3) // 4)

    def fwd(self, X, out, out_col, gi, gates, bcast=None, lengths=None):
        """X: Act [B,T,Cin]; out: Act whose columns [out_col, out_col+2H) receive h (fwd ++ bwd);
        gi: Act [B,T,6H] scratch; gates: raw tensor (T dtype, B*T*2*4H) or None.
        bcast = (vec fp32 [n, >= 2H], idx int64 [B], col): out[b, t, col:col+2H] = vec[idx[b], :2H] for every t (append_emb).
        lengths: int32 device tensor [B] (ragged batch, inference): the reverse direction of sample b starts at ITS last row --
        its gate inputs are reversed per sample over its own length, both directions run forward in time (ZsGruFwd.dir1_forward),
        and its outputs are reversed back.  Rows past a sample's length hold garbage."""
        c, H = self.ctx, self.H
        if lengths is not None and gates is not None:
            raise L.ZsError('GruLayer.fwd: ragged batches are an inference feature (no tape)')
        L.call('zs_gemm_conv', 'ZsGemmConv', c.stream, dtype=c.dt, A=X.ptr(), lda=X.ld, a_batch_stride=X.T * X.ld, B=X.B,
               T_in=X.T, T_out=X.T, taps=1, stride=1, pad_left=0, pad_mode=L.ZS_PAD_ZERO, gather=0, cin_pad=self.cin_pad,
               W=L.ptr(self.wih_f), ldw=self.ih_ldw, N=6 * H, n_pad=self.ih_npad, bias=L.ptr(self.bih), act=L.ZS_ACT_NONE,
               out=gi.ptr(), ldc=gi.ld, out_cols=min(gi.cols, rup(6 * H, 32)), groups=1)
        if lengths is not None:
            L.call('zs_rows_reverse', 'ZsRowsReverse', c.stream, dtype=c.dt, x=gi.ptr(), ld=gi.ld, col0=3 * H, cols=3 * H, B=X.B, T=X.T,
                   lengths=L.ptr(lengths))
        work = self._work(X.B)
        L.call('zs_gru_fwd', 'ZsGruFwd', c.stream, dtype=c.dt, B=X.B, T=X.T, H=H, gi=gi.ptr(), ldgi=gi.ld,
               whh=L.ptr(self.whh_f), ldw=self.hh_ldw, n_pad=self.hh_npad, w_gstride=self.hh_npad * self.hh_ldw,
               bhh=L.ptr(self.bhh), bhh_gstride=3 * H, out=out.ptr(), ldo=out.ld, out_col=out_col,
               gates=L.ptr(gates) if gates is not None else None, work=L.ptr(work), work_bytes=work.numel() * 4,
               whh_interleaved=int(self.fast), status=L.ptr(c.status),
               bcast_vec=(L.ptr(bcast[0]) if bcast else None), bcast_ld=(bcast[0].shape[1] if bcast else 0),
               bcast_idx=(L.ptr(bcast[1]) if bcast else None), bcast_col=(bcast[2] if bcast else 0),
               dir1_forward=int(lengths is not None))
        if lengths is not None:
            L.call('zs_rows_reverse', 'ZsRowsReverse', c.stream, dtype=c.dt, x=out.ptr(), ld=out.ld, col0=out_col + H, cols=H, B=X.B, T=X.T,
                   lengths=L.ptr(lengths))

    def check(self, B):
        """Test hook: raise if the last persistent pass over this layer's work buffer timed out (synchronises)."""
        L.check(L.lib().zs_gru_check(L.ptr(self._work(B)), B, self.H, self.ctx.stream), 'zs_gru_check')

    def bwd(self, dout, dout_col, out, out_col, gates, X, dgi, dgh, dX, add_src=None, colsum=None):
        """BPTT + parameter gradients + input gradient dX (= dgi W_ih, + add_src)."""
        c, H = self.ctx, self.H
        B, T = X.B, X.T
        work = self._work(B)
        L.call('zs_gru_bwd', 'ZsGruBwd', c.stream, dtype=c.dt, B=B, T=T, H=H, dout=dout.ptr(), ldd=dout.ld, dout_col=dout_col,
               out=out.ptr(), ldo=out.ld, out_col=out_col, gates=L.ptr(gates), whh_t=L.ptr(self.whh_t), ldw=self.hh_ldw_t,
               n_pad=self.hh_npad_t, w_gstride=self.hh_npad_t * self.hh_ldw_t, dgi=dgi.ptr(), ldgi=dgi.ld, dgh=dgh.ptr(),
               ldgh=dgh.ld, work=L.ptr(work), work_bytes=work.numel() * 4, status=L.ptr(c.status))
        for d in range(2):
            # dW_hh[d] = sum_t dgh_t^T h_{t-1}  (dir 0: h_{t-1} = out[t-1]; dir 1: out[t+1]) ; zero rows outside
            wgrad_call(c, dict(dtype=c.dt, dY=dgh.ptr(3 * H * d), ldy=dgh.ld, y_cols=3 * H, X=out.ptr(out_col + d * H),
                               ldx=out.ld, x_batch_stride=T * out.ld, x_cols=H, B=B, T_in=T, T_out=T, taps=1, stride=1,
                               pad_left=(1 if d == 0 else -1), pad_mode=L.ZS_PAD_ZERO, Cout=3 * H, Cin=H,
                               dW=L.ptr(self.gw_hh[d]), so=H, si=1, sj=0, db=L.ptr(self.gb_hh[d]), co_split2=0,
                               accumulate=0, splits=0), self.sids[2 * d])
            wgrad_call(c, dict(dtype=c.dt, dY=dgi.ptr(3 * H * d), ldy=dgi.ld, y_cols=3 * H, X=X.ptr(), ldx=X.ld,
                               x_batch_stride=T * X.ld, x_cols=min(X.cols, rup(self.Cin, 16 // c.es)), B=B, T_in=T, T_out=T,
                               taps=1, stride=1, pad_left=0, pad_mode=L.ZS_PAD_ZERO, Cout=3 * H, Cin=self.Cin,
                               dW=L.ptr(self.gw_ih[d]), so=self.Cin, si=1, sj=0, db=L.ptr(self.gb_ih[d]), co_split2=0,
                               accumulate=0, splits=0), self.sids[2 * d + 1])
        kw = dict(dtype=c.dt, A=dgi.ptr(), lda=dgi.ld, a_batch_stride=T * dgi.ld, B=B, T_in=T, T_out=T, taps=1, stride=1,
                  pad_left=0, pad_mode=L.ZS_PAD_ZERO, gather=0, cin_pad=self.g6_pad, W=L.ptr(self.wih_d), ldw=self.ih_ldw_d,
                  N=self.Cin, n_pad=self.ih_npad_d, act=L.ZS_ACT_NONE, out=dX.ptr(), ldc=dX.ld,
                  out_cols=min(dX.cols, rup(self.Cin, 32)), groups=1)
        if add_src is not None:
            kw.update(add_src=add_src.ptr(), add_ld=add_src.ld, add_f32=0)
        if colsum is not None:
            kw.update(colsum=colsum[0], colsum_ld=colsum[1], colsum_col0=colsum[2])
        L.call('zs_gemm_conv', 'ZsGemmConv', c.stream, **kw)
