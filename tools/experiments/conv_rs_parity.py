"""The parity test of tools/experiments/conv_rs.hip as it ran (4 / 4 green) while the kernel was in the library behind
dsg_set_tuning key 33 (see README.md here for how to put it back).  Kept as a record; not collected by pytest."""
@pytest.mark.parametrize("h,w,batch,res,temb", [(64, 64, 2, True, True), (128, 96, 1, False, True), (72, 64, 3, True, False), (256, 256, 1, True, True)])
def test_row_streaming_64_channel_conv(h, w, batch, res, temb):
    """csrc/conv_rs.hip (dsg_set_tuning key 33): the 64 -> 64 channel 3x3 conv with GroupNorm + SiLU in front as a row-streaming
    kernel -- the contraction split over the four waves by input channel, each wave's weight slice resident in registers, the four
    partial tiles added through LDS once per output row.  Against fp64 in the split convs' round-off class, against the
    conv_h2_kernel call of the same arguments (another summation order: fp32 round-off), the epilogue statistics of what was
    written, and row i of a batch bitwise the batch-1 call on row i."""
    from drivescenegen_amd import _lib
    d = lambda t: None if t is None else t.to(DEV)
    c = cout = 64
    x = _t(81, (batch, c, h, w), 1.4)
    wt = _t(82, (cout, c, 3, 3), 1.0 / np.sqrt(9 * c))
    bias = _t(83, (cout,), 0.1)
    gamma, beta = 1 + _t(84, (c,), 0.1), _t(85, (c,), 0.1)
    r = _t(86, (batch, cout, h, w)) if res else None
    tproj = _t(87, (batch, cout + 5), 0.5)
    a = F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5))
    want = F.conv2d(a, wt.double(), bias.double(), padding=1)
    mag = F.conv2d(a.abs(), wt.double().abs(), None, padding=1) + 1e-30
    if temb:
        want = want + tproj[:, 2:2 + cout, None, None].double()
    if res:
        want = want + r.double()
    ss = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-5)
    tp = d(tproj)
    kw = dict(ksize=3, cout=cout, src_blocked=True, dst_blocked=True, want_stats=True, gn_scale_shift=ss, silu=True,
              weight_h2=ops.relayout_conv_weight_h2(d(wt)), temb=tp[:, 2:] if temb else None, temb_stride=tp.stride(0))
    xb, wr = ops.to_blocked(d(x)), ops.relayout_conv_weight(d(wt))
    rb = ops.to_blocked(d(r)) if res else None
    lib = _lib.load()
    got = {}
    try:
        for on in (0, 1):
            _lib.check(lib.dsg_set_tuning(33, on))
            y, st = ops.conv2d_fused(xb, wr, d(bias), residual=rb, **kw)
            got[on] = (ops.from_blocked(y).cpu(), st)
        y1, _ = ops.conv2d_fused(xb[:1].contiguous(), wr, d(bias), residual=None if rb is None else rb[:1].contiguous(),
                                 **dict(kw, gn_scale_shift=ss[:1].contiguous(), temb=tp[:1, 2:] if temb else None))
    finally:
        lib.dsg_set_tuning(33, 0)
    assert not torch.equal(got[0][0], got[1][0])   # (another kernel really ran)
    e0 = float(((got[0][0].double() - want).abs() / mag).max())
    e1 = float(((got[1][0].double() - want).abs() / mag).max())
    assert e1 <= max(2 * e0, 6e-7), (e1, e0)
    assert float((got[0][0].double() - got[1][0].double()).abs().max()) <= 4e-6 * float(want.abs().max())
    y, st = got[1]
    assert st is not None and st.shape == got[0][1].shape
    ref = torch.stack([y.double().sum((2, 3)), (y.double() ** 2).sum((2, 3))], -1)
    assert torch.allclose(st.sum(2).cpu(), ref, rtol=3e-6, atol=1e-4)
    assert torch.equal(ops.from_blocked(y1).cpu()[0], y[0])


