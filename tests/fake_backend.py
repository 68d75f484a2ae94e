"""TEST DOUBLE for coclr_amd.ops: every kernel entry point re-expressed with ATen CPU ops
on the same buffers/layouts (packed weights, [2][C][ntiles] statistics, channel-slice
views, int32 pool indices, device-side queue pointer).

It exists so the HOST logic -- the engine's tape, concat-free inception wiring, gradient
accumulation, the InfoNCE/UberNCE/CoCLR step sequencing and the gloo/world-size-2
collectives -- can be exercised in the CPU-only test tier.  It is installed only by
tests (`install(monkeypatch)`); the product has no CPU path and never imports this file.
"""
import torch
import torch.nn.functional as F

from coclr_amd import ops


def _r32(v):
    return (v + 31) // 32 * 32


def _r128(v):
    return (v + 127) // 128 * 128


def conv_packed_size(cin, cout, taps, transpose):
    r, c = (cout, cin) if int(transpose) & 1 else (cin, cout)
    return taps * _r32(r) * _r128(c)


def conv_pack_weights(w, packed, cout, cin, taps, co_stride, ci_stride, tap_base, transpose,
                      tap_step=1, row0=0, rows_total=0, col0=0, cols_total=0):
    wino = bool(int(transpose) & 2)
    transpose = bool(int(transpose) & 1)
    if wino:
        # the double keeps the plain stencil (3, 7 or 9 taps) in the first slots of the 4 / 6 / 9 / 16-slot operand
        real = {4: 3, 5: 4, 6: 3, 9: 7, 16: 9}[taps]
        w3 = torch.as_strided(w.detach().reshape(-1), (cout, cin, real), (co_stride, ci_stride, tap_step), tap_base)
        r, c = (cout, cin) if transpose else (cin, cout)
        dst = packed.view(taps, _r32(r), _r128(c))
        dst.zero_()
        src = w3.flip(2).permute(2, 0, 1) if transpose else w3.permute(2, 1, 0)
        dst[:real, :r, :c] = src
        return
    w3 = torch.as_strided(w.detach().reshape(-1), (cout, cin, taps),
                          (co_stride, ci_stride, tap_step), tap_base)
    r, c = (cout, cin) if transpose else (cin, cout)
    placed = rows_total > 0 and cols_total > 0
    dst = packed.view(taps, _r32(rows_total if placed else r), _r128(cols_total if placed else c))
    if not placed:
        dst.zero_()
    src = w3.flip(2).permute(2, 0, 1) if transpose else w3.permute(2, 1, 0)
    dst[:, row0:row0 + r, col0:col0 + c] = src


def conv_pack_describe(w, packed, cout, cin, taps, co_stride, ci_stride, tap_base, transpose,
                       tap_step=1, row0=0, rows_total=0, col0=0, cols_total=0):
    return [0] * 16, 1


def conv_pack_batch(table, blockmap, requests=None):
    for args, kw in requests:
        conv_pack_weights(*args, **kw)


def _unpack(geom, wp):
    taps = geom.taps
    if getattr(geom, "algo", 0) >= 1:
        slots = {3: 6 if geom.algo == 2 else 4, 4: 5, 7: 9, 9: 16}[taps]
        w = wp.view(slots, _r32(geom.Cin), _r128(geom.Cout))[:taps, :geom.Cin, :geom.Cout]
        return w.permute(2, 1, 0).reshape(geom.Cout, geom.Cin, *geom.k).contiguous()
    w = wp.view(taps, _r32(geom.Cin), _r128(geom.Cout))[:, :geom.Cin, :geom.Cout]
    return w.permute(2, 1, 0).reshape(geom.Cout, geom.Cin, *geom.k).contiguous()


def _virtual_input(geom, x, n_index):
    xs = x if n_index is None else x[n_index]
    xs = xs[:geom.N]
    d = geom.d
    if d != (1, 1, 1):
        T, H, W = xs.shape[2:]
        up = xs.new_zeros(xs.shape[0], xs.shape[1], (T - 1) * d[0] + 1, (H - 1) * d[1] + 1,
                          (W - 1) * d[2] + 1)
        up[:, :, ::d[0], ::d[1], ::d[2]] = xs
        xs = up
    return xs


def _conv_raw(geom, xs, w):
    """conv with possibly negative padding, output cropped / zero-extended to geom.odim."""
    pad = list(geom.p)
    sl = [slice(None)] * 3
    for i in range(3):
        if pad[i] < 0:
            sl[i] = slice(-pad[i], None)
            pad[i] = 0
    xs = xs[:, :, sl[0], sl[1], sl[2]]
    # make sure every requested output position exists: extend the input with zeros at the end
    need = [(geom.odim[i] - 1) * geom.s[i] + geom.k[i] - 2 * pad[i] for i in range(3)]
    extra = [max(0, need[i] - xs.shape[2 + i]) for i in range(3)]
    if any(extra):
        xs = F.pad(xs, (0, extra[2], 0, extra[1], 0, extra[0]))
    out = F.conv3d(xs, w, None, geom.s, tuple(pad))
    return out[:, :, :geom.odim[0], :geom.odim[1], :geom.odim[2]]


def conv_fwd(geom, x, w_packed, y, stats=None, bias=None, ep_scale=None, ep_shift=None,
             n_index=None, relu=False, accumulate=False):
    w = _unpack(geom, w_packed)
    raw = _conv_raw(geom, _virtual_input(geom, x, n_index), w)
    if getattr(geom, "lattice", None) is not None:
        ys, yo, _ = geom.lattice
        y = y[:, :, yo[0]::ys[0], yo[1]::ys[1], yo[2]::ys[2]]     # view: copy_ below lands in place
    if accumulate:
        raw = raw + y
    if stats is not None:
        st = stats.view(2, geom.Cout, geom.ntiles())
        st.zero_()
        st[0, :, 0] = raw.sum((0, 2, 3, 4))
        st[1, :, 0] = (raw * raw).sum((0, 2, 3, 4))
    v = raw
    if bias is not None:
        v = v + bias.view(1, -1, 1, 1, 1)
    if ep_scale is not None:
        v = v * ep_scale.view(1, -1, 1, 1, 1) + ep_shift.view(1, -1, 1, 1, 1)
    if relu:
        v = torch.relu(v)
    y.copy_(v)


def conv_fwd_multi(calls):
    """The pair / multi entry point: every problem on its own."""
    for c in calls:
        bb = c.get("bwd_bn")
        ia = c.get("in_affine")
        if ia is not None:
            # the raw output of a BatchNorm unit, normalised while the convolution reads it
            xa = c["x"] * _b(ia[0]) + _b(ia[1])
            c = dict(c, x=torch.relu(xa) if ia[2] else xa)
        conv_fwd(c["geom"], c["x"], c["w"], c["y"], stats=None if bb is not None else c.get("stats"),
                 n_index=c.get("n_index"), accumulate=c.get("accumulate", False))
        if bb is not None:
            # the data gradient that writes a BatchNorm unit's dz also forms that unit's backward sums
            geom, y = c["geom"], c["y"]
            assert not c.get("accumulate", False) and c.get("stats") is not None and geom.bwd_sums_ok()
            by, scale, shift, mean, invstd, relu = bb
            assert by.shape == y.shape
            sel = (slice(None),) * 5
            if getattr(geom, "lattice", None) is not None:
                ys, yo, _ = geom.lattice
                sel = (slice(None), slice(None), slice(yo[0], None, ys[0]), slice(yo[1], None, ys[1]),
                       slice(yo[2], None, ys[2]))
            g, u = y[sel], by[sel]
            if relu:
                g = g * ((u * _b(scale) + _b(shift)) > 0)
            st = c["stats"].view(2, geom.Cout, geom.ntiles())
            st.zero_()
            st[0, :, 0] = g.sum((0, 2, 3, 4))
            st[1, :, 0] = (g * ((u - _b(mean)) * _b(invstd))).sum((0, 2, 3, 4))


def bn_finalize_apply_multi(units):
    for u in units:
        gamma, beta, rm, rv, nbt, momentum, eps = u["bn"]
        mean, invstd, scale, shift = u["small"]
        bn_finalize_apply(u["stats"], u["C"], u["ntiles"], u["count"], gamma, beta, rm, rv, nbt, momentum,
                          eps, mean, invstd, scale, shift, u["y"], u["z"], u["relu"], c0=u.get("c0", 0),
                          c_total=u.get("c_total"))


def bn_act_backward_multi(units):
    for u in units:
        sums = None
        if u.get("partials"):
            # the reduction was done by the data gradient(s) that wrote dz: fold their partials
            C_ = u["y"].shape[1]
            sg = sum(st.view(2, C_, nt)[0].double().sum(1) for st, nt in u["partials"])
            sgx = sum(st.view(2, C_, nt)[1].double().sum(1) for st, nt in u["partials"])
            sums = (sg, sgx)
        bn_act_backward(u["dz"], u["y"], None, u["scale"], u["shift"], u["mean"], u["invstd"], u["sums"],
                        u["dy"], None, u["dgamma"], u["dbeta"], u["relu"], u["training"], _sums=sums)


def conv_wgrad(geom, x, dy, dw, workspace, co_stride, ci_stride, tap_base, accumulate=False):
    w0 = torch.zeros(geom.Cout, geom.Cin, *geom.k, requires_grad=True)
    with torch.enable_grad():
        out = _conv_raw(geom, _virtual_input(geom, x.detach(), None), w0)
        out.backward(dy.detach()[:geom.N])
    g = w0.grad.reshape(geom.Cout, geom.Cin, geom.taps)
    r0 = 0
    for t in (dw if isinstance(dw, (list, tuple)) else [dw]):
        rows = t.shape[0] if isinstance(dw, (list, tuple)) else geom.Cout
        dst = torch.as_strided(t, (rows, geom.Cin, geom.taps), (co_stride, ci_stride, 1),
                               t.storage_offset() + tap_base)
        if accumulate:
            dst.add_(g[r0:r0 + rows])
        else:
            dst.copy_(g[r0:r0 + rows])
        r0 += rows


def bn_finalize(stats, C_, ntiles, count, gamma, beta, running_mean, running_var, nbt, momentum,
                eps, mean, invstd, scale, shift, c0=0, c_total=None):
    c_total = C_ if c_total is None else c_total
    st = stats.view(2, c_total, ntiles)[:, c0:c0 + C_].double().sum(-1)
    mu = st[0] / count
    var = (st[1] / count - mu * mu).clamp_min(0)
    inv = 1.0 / torch.sqrt(var + eps)
    mean.copy_(mu.float())
    invstd.copy_(inv.float())
    scale.copy_(gamma * invstd)
    shift.copy_(beta - mean * scale)
    if running_mean is not None:
        unb = var * count / (count - 1) if count > 1 else var
        running_mean.mul_(1 - momentum).add_(momentum * mu.float())
        running_var.mul_(1 - momentum).add_(momentum * unb.float())
    if nbt is not None:
        nbt += 1


def bn_eval_affine(gamma, beta, running_mean, running_var, eps, C_, mean, invstd, scale, shift):
    mean.copy_(running_mean)
    invstd.copy_(1.0 / torch.sqrt(running_var + eps))
    scale.copy_(gamma * invstd)
    shift.copy_(beta - running_mean * scale)


def _b(v):
    return v.view(1, -1, 1, 1, 1)


def bn_finalize_apply(stats, C_, ntiles, count, gamma, beta, running_mean, running_var, nbt,
                      momentum, eps, mean, invstd, scale, shift, y, z, relu, c0=0, c_total=None):
    bn_finalize(stats, C_, ntiles, count, gamma, beta, running_mean, running_var, nbt, momentum, eps,
                mean, invstd, scale, shift, c0=c0, c_total=c_total)
    bn_act_apply(y, scale, shift, None, z, relu)


SMALL_CHANNEL = 32768


def bn_act_apply(y, scale, shift, residual, z, relu):
    v = y * _b(scale) + _b(shift)
    if residual is not None:
        v = v + residual
    z.copy_(torch.relu(v) if relu else v)


def bn_backward_workspace(N, C_):
    return 2 * C_ * N


def bn_act_backward(dz, y, z, scale, shift, mean, invstd, sums_ws, dy, dres, dgamma,
                    dbeta, relu, training, dres_accumulate=False, _sums=None):
    g = dz
    if relu:
        mask = (z > 0) if z is not None else ((y * _b(scale) + _b(shift)) > 0)
        g = dz * mask
    xhat = (y - _b(mean)) * _b(invstd)
    if _sums is not None:
        sg, sgx = _sums
    else:
        sg = g.double().sum((0, 2, 3, 4))
        sgx = (g.double() * xhat.double()).sum((0, 2, 3, 4))
    if dgamma is not None:
        dgamma.copy_(sgx.float())
    if dbeta is not None:
        dbeta.copy_(sg.float())
    if dres is not None:
        if dres_accumulate:
            dres.add_(g)
        else:
            dres.copy_(g)
    if training:
        cnt = y.numel() / y.shape[1]
        dy.copy_(_b(scale) * (g - _b((sg / cnt).float()) - xhat * _b((sgx / cnt).float())))
    else:
        dy.copy_(_b(scale) * g)


def bn_act_backward_coeffs(dz, y, scale, shift, mean, invstd, sums_ws, coef, dgamma, dbeta, relu, training):
    g = dz * ((y * _b(scale) + _b(shift)) > 0) if relu else dz
    xhat = (y - _b(mean)) * _b(invstd)
    sg = g.double().sum((0, 2, 3, 4))
    sgx = (g.double() * xhat.double()).sum((0, 2, 3, 4))
    if dgamma is not None:
        dgamma.copy_(sgx.float())
    if dbeta is not None:
        dbeta.copy_(sg.float())
    c = coef.view(5, -1)
    c[0].copy_(scale)
    c[3].copy_(scale)
    c[4].copy_(shift)
    if training:
        cnt = y.numel() / y.shape[1]
        mg, mgx = (sg / cnt).float(), (sgx / cnt).float()
        c[1].copy_(-scale * invstd * mgx)
        c[2].copy_(scale * (mean * invstd * mgx - mg))
    else:
        c[1].zero_()
        c[2].zero_()


def conv_wgrad_bn(geom, x, dz, y, coef, relu, dw, workspace, co_stride, ci_stride, tap_base=0,
                  accumulate=False):
    c = coef.view(5, -1)
    g = dz * ((y * _b(c[3]) + _b(c[4])) > 0) if relu else dz
    conv_wgrad(geom, x, _b(c[0]) * g + _b(c[1]) * y + _b(c[2]), dw, workspace, co_stride, ci_stride, tap_base,
               accumulate)


def maxpool_fwd(geom, x, y, indices=None, in_scale=None, in_shift=None, in_relu=False):
    if in_scale is not None:
        x = x * _b(in_scale) + _b(in_shift)
        if in_relu:
            x = torch.relu(x)
    out, idx = F.max_pool3d(x, geom.k, geom.s, geom.p, return_indices=True)
    y.copy_(out)
    if indices is not None:
        indices.copy_(idx.to(torch.int32))


def maxpool_bwd(geom, dy, indices, dx, accumulate=False):
    N, Cc = dx.shape[:2]
    flat = torch.zeros(N, Cc, dx.shape[2] * dx.shape[3] * dx.shape[4])
    flat.scatter_add_(2, indices.reshape(N, Cc, -1).long(), dy.reshape(N, Cc, -1))
    g = flat.view(N, Cc, *dx.shape[2:])
    if accumulate:
        dx.add_(g)
    else:
        dx.copy_(g)


def bn_act_backward_pooled(geom, pool_dy, indices, y, scale, shift, mean, invstd, sums_ws, dy, dgamma,
                           dbeta, relu, training):
    dz = torch.empty_like(y)
    maxpool_bwd(geom, pool_dy, indices, dz)
    bn_act_backward(dz, y, None, scale, shift, mean, invstd, sums_ws, dy, None, dgamma, dbeta, relu,
                    training)


def pooled_backward_fits(geom):
    return True


def global_avgpool_fwd(x, y):
    y.copy_(x.mean((2, 3, 4), keepdim=True))


def global_avgpool_bwd(dy, dx):
    S = dx.shape[2] * dx.shape[3] * dx.shape[4]
    dx.copy_((dy / S).expand_as(dx))


def gemm_workspace(M, N, K, splits):
    return M * N * splits if splits > 1 else 0


def gemm(a, sam, sak, b, sbk, sbn, c, ldc, bias, M, N, K, alpha=1.0, relu=False, accumulate=False,
         splits=1, workspace=None):
    A = torch.as_strided(a, (M, K), (sam, sak))
    Bm = torch.as_strided(b, (K, N), (sbk, sbn))
    v = alpha * (A @ Bm)
    if bias is not None:
        v = v + bias
    if relu:
        v = torch.relu(v)
    dst = torch.as_strided(c, (M, N), (ldc, 1))
    if accumulate:
        dst.add_(v)
    else:
        dst.copy_(v)


def gemm_fused_workspace(M, N, K, splits):
    return M * N * max(1, splits)


def gemm_fused(a, sam, sak, b, sbk, sbn, c, ldc, bias, M, N, K, alpha=1.0, relu=False, splits=1,
               workspace=None, mode=0, S=0, ep_a=None, lda=0, ep_b=None, ep_y=None, inv_norm=None, out2=None,
               f=0.0, rowsum=None):
    A = torch.as_strided(a, (M, K), (sam, sak))
    Bm = torch.as_strided(b, (K, N), (sbk, sbn))
    v = alpha * (A @ Bm)
    if bias is not None:
        v = v + bias
    if relu:
        v = torch.relu(v)
    if rowsum is not None:
        rowsum.reshape(-1).copy_(A.sum(1))
    if mode == 1:
        v = v * (torch.as_strided(ep_a, (M, N), (lda, 1)) > 0)
    elif mode == 2:
        inv = 1.0 / v.norm(dim=1).clamp_min(f)
        if out2 is not None:
            out2.copy_(inv)
        v = v * inv[:, None]
    elif mode == 3:
        v = v + torch.as_strided(ep_a, (M, 1), (lda, 1)) * f * ep_b
        dot = (v * ep_y).sum(1, keepdim=True)
        v = (v - ep_y * dot) * inv_norm[:, None]
    elif mode == 4:
        c.reshape(M, N, S).copy_((v / S)[:, :, None].expand(M, N, S))
        return
    c.reshape(-1)[:0]          # (c may be a 5-d parameter-shaped gradient: written through its flat memory)
    torch.as_strided(c, (M, N), (ldc, 1)).copy_(v)


def l2norm_fwd(x, y, inv_norm, eps=1e-12):
    inv = 1.0 / x.norm(dim=1).clamp_min(eps)
    y.copy_(x * inv[:, None])
    if inv_norm is not None:
        inv_norm.copy_(inv)


def l2norm_bwd(dy, y, inv_norm, dx):
    dot = (dy * y).sum(1, keepdim=True)
    dx.copy_((dy - y * dot) * inv_norm[:, None])


def nce_logits_fwd(q, k, queue, logits, T):
    logits[:, 0] = (q * k).sum(1) / T
    logits[:, 1:] = (q @ queue) / T


def nce_logits_bwd(dlogits, k, queue, dq, workspace, T, splits):
    dq.copy_((dlogits[:, 1:] @ queue.t() + dlogits[:, :1] * k) / T)


def momentum_update(table, nchunks, m, one_minus_m, pairs=None):
    mf = torch.tensor(m, dtype=torch.float32)
    of = torch.tensor(one_minus_m, dtype=torch.float32)
    for dst, src in zip(*pairs):
        dst.data.copy_(dst.data * mf + src.data * of)


def queue_enqueue(queue, keys, ptr):
    p = int(ptr)
    queue[:, p:p + keys.shape[0]] = keys.t()


def queue_fill_i64(queue, vals, const_val, BW, ptr):
    p = int(ptr)
    queue[p:p + BW] = vals if vals is not None else const_val


def queue_advance(ptr, BW, K):
    ptr[0] = (int(ptr) + BW) % K


def positive_mask(sim, src, names, mask, topk):
    same = src[:, None] == names[None, :]
    m = same.clone()
    if topk > 0:
        s = sim.clone()
        s[same] = -float("inf")
        _, idx = torch.topk(s, topk, dim=1)
        m.scatter_(1, idx, True)
    mask[:, 0] = 1
    mask[:, 1:] = m.to(mask.dtype)


def mine_workspace(B, K, topk, device):
    return (torch.empty(1), torch.empty(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32))


def mine_positives(kf, queue_second, src, names, mask, topk, workspace, sim_out=None):
    sim = kf.matmul(queue_second)
    if sim_out is not None:
        sim_out.copy_(sim)
    positive_mask(sim, src, names, mask, topk)


def gather_rows(inp, idx, out):
    out.copy_(inp[idx])


def pull_rows(row_ptrs, out, keep=None):
    raise NotImplementedError("the peer row pull maps device memory across processes: GPU tier only")


def relu_fwd(x, y):
    y.copy_(torch.relu(x))


def relu_bwd(dy, y, dx):
    dx.copy_(dy * (y > 0))


def colsum(x, out):
    out.copy_(x.sum(0))


def sigmoid_fwd(s_, w):
    w.copy_(torch.sigmoid(s_))


def sigmoid_bwd(dw, w, ds):
    ds.copy_(dw * w * (1 - w))


def plane_scale(a, gain, bias, out, accumulate=False):
    v = a * gain[:, :, None, None, None]
    if bias is not None:
        v = v + bias[:, :, None, None, None]
    if accumulate:
        out.add_(v)
    else:
        out.copy_(v)


def plane_dot(a, b, out):
    out.copy_((a * b).sum((2, 3, 4)))


# ---- training-loop neighbours / evaluation consumers (host-logic doubles) -----------------

def nce_loss_fwd(logits, mask, target, rowstats, flags, scalars, mode, drop_self=False, k1=1, k2=5):
    B, N1 = logits.shape
    lg = logits.detach().double()
    lse = torch.logsumexp(lg, 1)
    if mode == 0:
        pos = torch.zeros(B, N1, dtype=torch.bool)
        pos[torch.arange(B), target] = True
    else:
        pos = mask.bool()
    eff = pos.clone()
    drop = torch.zeros(B, dtype=torch.bool)
    if mode == 1 and drop_self:
        drop = (pos.sum(1) != 1) & pos[:, 0]
        eff[drop, 0] = False
    neg_inf = torch.full_like(lg, -float("inf"))
    if mode == 2:
        n = eff.sum(1).double()
        loss = lse - (lg * eff).sum(1) / n
        aux = n
    else:
        lsp = torch.logsumexp(torch.where(eff, lg, neg_inf), 1)
        loss = lse - lsp
        aux = lsp
    pmax = torch.where(pos, lg, neg_inf).max(1).values
    cgp = (lg > pmax[:, None]).sum(1)
    cg0 = (lg > lg[:, :1]).sum(1)
    rowstats[:, 0] = loss.float(); rowstats[:, 1] = lse.float(); rowstats[:, 2] = aux.float()
    rowstats[:, 3] = (cgp < k1).float(); rowstats[:, 4] = (cgp < k2).float()
    rowstats[:, 5] = (cg0 < k1).float(); rowstats[:, 6] = (cg0 < k2).float()
    rowstats[:, 7] = lg.max(1).values.float()
    flags.copy_(drop.to(torch.uint8))
    scalars.copy_(rowstats[:, [0, 3, 4, 5, 6]].double().mean(0).float())


def nce_loss_bwd(logits, mask, target, rowstats, flags, dloss, dlogits, mode):
    B, N1 = logits.shape
    lg = logits.detach()
    sm = torch.exp(lg - rowstats[:, 1:2])
    if mode == 0:
        w = torch.zeros_like(lg)
        w[torch.arange(B), target] = 1.0
    else:
        eff = mask.bool().clone()
        eff[flags.bool(), 0] = False
        w = torch.exp(lg - rowstats[:, 2:3]) * eff if mode == 1 else eff.float() / rowstats[:, 2:3]
    dlogits.copy_((sm - w) * (dloss.reshape(()) / B))


def stage_clips(frames, out, S, mean, std):
    B, C_ = frames.shape[0], frames.shape[1]
    x = frames.to(torch.float32) / 255 if frames.dtype == torch.uint8 else frames
    m = torch.tensor(list(mean), dtype=torch.float32).view(1, C_, 1, 1, 1)
    s = torch.tensor(list(std), dtype=torch.float32).view(1, C_, 1, 1, 1)
    x = (x - m) / s
    T = frames.shape[2] // S
    out.copy_(x.view(B, C_, S, T, *frames.shape[3:]).transpose(1, 2))


def colstats_workspace(rows, cols):
    return 2 * min(rows, 64) * cols


def bn1d_stats(x, stats, workspace):
    C_ = x.shape[1]
    stats[:C_] = x.double().sum(0).float()
    stats[C_:] = (x.double() ** 2).sum(0).float()


def center_rows(x, out, workspace):
    out.copy_(x - x.mean(0, keepdim=True))


def retrieval_hits(sim, train_label, test_label, ks, hits, topidx=None):
    kmax = topidx.shape[1]
    _, idx = torch.sort(sim, dim=1, descending=True, stable=True)
    idx = idx[:, :kmax]
    topidx.copy_(idx.to(torch.int32))
    match = train_label[idx] == test_label[:, None]
    for i, k in enumerate(ks.tolist()):
        hits[:, i] = match[:, :k].any(1).float()


_NAMES = [n for n, v in list(globals().items())
          if callable(v) and not n.startswith("_") and hasattr(ops, n) and n not in ("F",)]


def install(monkeypatch):
    for n in _NAMES:
        monkeypatch.setattr(ops, n, globals()[n])
