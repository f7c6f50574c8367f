"""VERDICT r04 item 1, conditions (b) and (c): the split-bf16 form of every GEMM family against a float64 reference, at the
family's bench shape (B = 256), next to the FP32-MFMA path on the same operands.
  (b) max and mean |error| <= 1.25 x the FP32-MFMA path's, |signed mean error| <= max(2 x the f32 path's, 1e-9 max|ref|)
      (the bf16 MFMA's adder truncates toward -inf; the (plain, negated) accumulator pairs must cancel that bias);
  (c) range and specials: operands scaled by 2^+-100 keep the same accuracy; rows holding Inf / NaN give non-finite outputs
      exactly where the f32 path does and leave every other row untouched.
Reference arithmetic: /root/reference/core/networks.py:66-81 under core/ddpg.py:136-139 (float32 conv / BatchNorm / ReLU)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cases():
    from tests import split_cases as sc
    return {
        "fwd_wide.sa2.l2": lambda: sc.FwdWide(27240, 128, 128, "act"),
        "fwd_wide.sa2.l3": lambda: sc.FwdWide(27240, 128, 256, "pool"),
        "fwd_wide.sa3.l2": lambda: sc.FwdWide(8192, 256, 256, "act"),
        "fwd_wide.sa3.l3": lambda: sc.FwdWide(8192, 256, 512, "pool"),
        "fwd_wide.sa2.l1": lambda: sc.FwdWide(27240, 128, 128, "gather"),
        "fwd_wide.sa3.l1": lambda: sc.FwdWide(8192, 256, 256, "gather"),
        "fwd_stream.sa1.l2": lambda: sc.FwdStream(213034, 64, 64, "act"),
        "fwd_stream.sa1.l3": lambda: sc.FwdStream(213032, 64, 128, "pool"),
        "dx_wide.sa2.l3": lambda: sc.DxWide(27240, 256, 128, "pool"),
        "dx_wide.sa2.l2": lambda: sc.DxWide(27240, 128, 128, "act"),
        "dx_wide.sa2.l1": lambda: sc.DxWide(27240, 128, 128, "scatter"),
        "dx_wide.sa3.l3": lambda: sc.DxWide(8192, 512, 256, "pool"),
        "dx_wide.sa3.l2": lambda: sc.DxWide(8192, 256, 256, "act"),
        "dx_wide.sa3.l1": lambda: sc.DxWide(8192, 256, 256, "scatter"),
        "dw_wide.sa2.l3": lambda: sc.DwWide(27240, 256, 128, "pool"),
        "dw_wide.sa2.l2": lambda: sc.DwWide(27240, 128, 128, "act"),
        "dw_wide.sa2.l1": lambda: sc.DwWide(27240, 128, 128, "gather"),
        "dw_wide.sa3.l3": lambda: sc.DwWide(8192, 512, 256, "pool"),
        "dw_wide.sa3.l2": lambda: sc.DwWide(8192, 256, 256, "act"),
        "dw_wide.sa3.l1": lambda: sc.DwWide(8192, 256, 256, "gather"),
        "bwd_stream.sa1.l3": lambda: sc.BwdStream(213034, 128, "pool"),
        "bwd_stream.sa1.l2": lambda: sc.BwdStream(213034, 64, "act"),
        "dx_stream.sa1.l3": lambda: sc.BwdStream(213034, 128, "pool", fused=False),
        "dx_stream.sa1.l2": lambda: sc.BwdStream(213034, 64, "act", fused=False),
    }


CASES = ("fwd_stream.sa1.l2", "fwd_stream.sa1.l3", "fwd_wide.sa2.l2", "fwd_wide.sa2.l3", "fwd_wide.sa3.l2", "fwd_wide.sa3.l3", "fwd_wide.sa2.l1", "fwd_wide.sa3.l1",
         "dx_wide.sa2.l3", "dx_wide.sa2.l2", "dx_wide.sa2.l1", "dx_wide.sa3.l3", "dx_wide.sa3.l2", "dx_wide.sa3.l1",
         "dw_wide.sa2.l3", "dw_wide.sa2.l2", "dw_wide.sa2.l1", "dw_wide.sa3.l3", "dw_wide.sa3.l2", "dw_wide.sa3.l1",
         "bwd_stream.sa1.l3", "bwd_stream.sa1.l2", "dx_stream.sa1.l3", "dx_stream.sa1.l2")


def check_case(case, name, report=None, keys=None, factor=1.25):
    from tests import split_cases as sc
    ref = case.ref()
    f32, r32 = case.run_mode(False)
    spl, rsp = case.run_mode(True)
    assert "split" in rsp and "split" not in r32, "%s: routed to %s / %s" % (name, r32, rsp)
    bad = []
    for k, rv in ref.items():
        if k.endswith("#abs") or (keys is not None and k not in keys):
            continue
        a32, m32, s32, scale = sc.errors(f32[k], rv)
        asp, msp, ssp, _ = sc.errors(spl[k], rv)
        if report is not None:
            report.append("%-18s %-9s f32-MFMA: max %.3e mean %.3e signed %+.3e | split: max %.3e mean %.3e signed %+.3e | of max|ref| %.3e"
                          % (name, k, a32, m32, s32, asp, msp, ssp, scale))
        # (statistics / gradient sums: f64 accumulation of f32 partial sums -- the same bounds apply; floors for outputs
        # whose f32 error happens to be ~0)
        # floor: 1e-9 of max |z| for a GEMM output.  A REDUCED output (BatchNorm statistics, dbeta / dgamma, scatter sums, dW:
        # column sums over up to 2e5 rows, accumulated in f32 per lane by BOTH paths) adds a per-element bias coherently -- the
        # split form's residual bias of ~2e-11 max |z| per element (the (plain, negated) accumulator pairs cancel the bf16
        # adder's truncation only to first order) becomes ~1e-9 of the sum of |terms| -- so its floor is 4e-9 of that sum
        floor = 4e-9 * float(ref[k + "#abs"].max()) if k + "#abs" in ref else 1e-9 * scale
        if asp > factor * a32 + floor or msp > factor * m32 + floor:
            bad.append("%s %s: split max %.3e mean %.3e vs f32 max %.3e mean %.3e" % (name, k, asp, msp, a32, m32))
        if abs(ssp) > max(2.0 * abs(s32), floor):
            bad.append("%s %s: split signed mean %.3e vs f32 %.3e (floor %.1e)" % (name, k, ssp, s32, floor))
    if "key" in f32:                                  # fused max-pool: keys decode to the launch's own raw output
        for out in (f32, spl):
            z = out["z"]
            key = out["key"][: z.shape[0] // case.gsz]
            hi = (key >> 32) & 0xffffffff
            bits = torch.where((hi & 0x80000000) != 0, hi ^ 0x80000000, (~hi) & 0xffffffff).to(torch.int32)
            val = bits.view(torch.float32)
            want = z[: key.shape[0] * case.gsz].view(-1, case.gsz, z.shape[1]).max(1).values
            assert torch.equal(val, want), name + ": pooled keys do not decode to the maxima of the stored output"
    return bad


@pytest.mark.parametrize("name", CASES)
def test_split_error_vs_float64_at_bench_shape(name):
    case = _cases()[name]()
    bad = check_case(case, name)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("name", ["fwd_wide.sa2.l2", "fwd_wide.sa2.l1"])
@pytest.mark.parametrize("exp", [100, -100])
def test_split_range(name, exp):
    """operands scaled by 2^+-100 (activations) and 2^-+100 (weights) / both by 2^+-50: same relative accuracy -- no term of the
    split overflows or is flushed where the f32 product survives"""
    from tests import split_cases as sc
    kind = name.split(".")[0]
    for a_s, w_s in ((2.0 ** exp, 0.05 * 2.0 ** -exp), (2.0 ** (exp // 2), 0.05 * 2.0 ** (exp // 2))):
        case = sc.FwdWide(4096, 128, 128, "gather" if name.endswith("l1") else "act", a_scale=a_s, w_scale=w_s)
        bad = check_case(case, "%s x2^%d" % (name, exp), keys=("z",))          # (the f32 statistics sums over- / underflow alike)
        assert not bad, "\n".join(bad)


@pytest.mark.parametrize("name", ["fwd_wide.sa2.l2"])
def test_split_specials(name):
    """rows holding +Inf / NaN / -Inf: outputs are non-finite exactly where the f32 path's are (Inf may become NaN: hi = Inf
    leaves mid = Inf - Inf; a NaN or -Inf pre-activation is squashed to 0 by the ReLU's max in both paths), every other row
    keeps its accuracy"""
    from tests import split_cases as sc
    case = sc.FwdWide(4096, 128, 128, "act")
    case.zin[17, 5] = float("inf")
    case.zin[99, 64] = float("nan")
    case.zin[200, 3] = -float("inf")           # relu(-inf * s + t) = 0 for s > 0: stays finite in both
    f32, _ = case.run_mode(False)
    spl, _ = case.run_mode(True)
    assert torch.equal(torch.isfinite(f32["z"]), torch.isfinite(spl["z"]))
    assert not torch.isfinite(f32["z"][17]).any() and torch.isfinite(f32["z"][200]).all()
    ref = case.ref()["z"]
    ok = torch.isfinite(f32["z"]).all(1) & torch.isfinite(ref).all(1)       # (the reference keeps the NaN the kernels' max squashes)
    e32 = (f32["z"].double() - ref)[ok].abs().max()
    esp = (spl["z"].double() - ref)[ok].abs().max()
    assert float(esp) <= 1.25 * float(e32) + 1e-12


EDGE = {
    # live rows that end inside a tile / slab, one row more than a tile, far fewer live rows than the static bound (grid-stride loops,
    # bounded buffer descriptors), the smallest shapes the routes accept
    "fwd_wide ragged": lambda sc: sc.FwdWide(2049, 128, 128, "act"),
    "fwd_wide ragged pool": lambda sc: sc.FwdWide(5004, 256, 512, "pool"),
    "fwd_wide ragged gather": lambda sc: sc.FwdWide(4097, 128, 256, "gather"),
    "fwd_stream ragged": lambda sc: sc.FwdStream(32768 + 37, 64, 64, "act"),
    "fwd_stream ragged pool": lambda sc: sc.FwdStream(32768 + 36, 64, 128, "pool"),
    "dx_wide ragged": lambda sc: sc.DxWide(2050, 128, 128, "act"),
    "dx_wide ragged pool": lambda sc: sc.DxWide(3001, 256, 128, "pool"),
    "dx_wide ragged scatter": lambda sc: sc.DxWide(2113, 128, 128, "scatter"),
    "dw_wide ragged": lambda sc: sc.DwWide(2100, 128, 128, "act"),
    "dw_wide ragged pool": lambda sc: sc.DwWide(4099, 512, 256, "pool"),
    "dw_wide ragged gather": lambda sc: sc.DwWide(2051, 256, 256, "gather"),
    "bwd_stream ragged": lambda sc: sc.BwdStream(32768 + 67, 64, "act"),
    "bwd_stream ragged pool": lambda sc: sc.BwdStream(32768 + 129, 128, "pool"),
    "dx_stream ragged pool": lambda sc: sc.BwdStream(32768 + 1, 128, "pool", fused=False),
    # an ODD number of K-tiles (the K loops are unrolled by two; every second block of 16 along the reduction index is negated, so
    # the last K-tile's pair of blocks must still cancel): K = 96 / 160 forward, n = 96 / 160 for the dX
    "fwd_wide odd ktiles": lambda sc: sc.FwdWide(4101, 96, 128, "act"),
    "fwd_wide odd ktiles pool": lambda sc: sc.FwdWide(4100, 160, 256, "pool"),
    "dx_wide odd ktiles": lambda sc: sc.DxWide(3003, 96, 128, "act"),
    "dx_wide odd ktiles pool": lambda sc: sc.DxWide(3001, 160, 128, "pool"),
}


@pytest.mark.parametrize("name", sorted(EDGE))
def test_split_edge_shapes(name):
    """tile / slab / quad boundaries: the split kernels against float64 where the live rows end inside a tile (rows past the live count
    must contribute nothing and stay unwritten: the output buffers are NaN-filled beyond them).  Fewer rows than at the bench
    shapes make the max-error ratio noisier: factor 2 instead of 1.25."""
    from tests import split_cases as sc
    case = EDGE[name](sc)
    bad = check_case(case, name, factor=2.0)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("kind", ["fwd_wide", "dx_wide", "dw_wide", "bwd_stream"])
def test_split_zero_live_rows(kind):
    """a device-side live-row count of 0 (an empty minibatch shard): no output is written, the statistics stay zero, nothing faults"""
    from tests import split_cases as sc
    case = {"fwd_wide": lambda: sc.FwdWide(4096, 128, 128, "act"), "dx_wide": lambda: sc.DxWide(4096, 128, 128, "act"),
            "dw_wide": lambda: sc.DwWide(4096, 128, 128, "act"), "bwd_stream": lambda: sc.BwdStream(40000, 64, "act")}[kind]()
    inner = getattr(case, "dx", case)
    inner.nrows.zero_()
    out, routed = case.run_mode(True)
    assert "split" in routed
    for k, v in out.items():
        if k in ("z", "gout"):
            continue                              # (row tensors: sliced by the host-side row count, NaN-filled, never written)
        assert float(v.double().abs().max()) == 0.0, k
    for k in ("z", "gout"):
        if k in out:
            assert bool(torch.isnan(out[k]).all()), k + " was written for dead rows"


TWO_GIB = {
    "fwd_stream": lambda sc: sc.FwdStream(4194304, 64, 128, "act", ragged=False),
    "fwd_wide": lambda sc: sc.FwdWide(2097152, 128, 256, "act", ragged=False),
    "dx_wide": lambda sc: sc.DxWide(2097152, 256, 128, "act", slack=0),
    "bwd_stream": lambda sc: sc.BwdStream(4194304, 128, "act", slack=0),
    "dx_stream": lambda sc: sc.BwdStream(4194304, 128, "act", fused=False, slack=0),
}


@pytest.mark.parametrize("kind", sorted(TWO_GIB))
def test_row_tensor_of_exactly_two_gib(kind):
    """the specialised routes address their row tensors through buffer descriptors with 32-bit unsigned byte offsets; the route
    predicates admit rows x pitch x 4 <= 2^31 (configs[3] through the facade: capacity 128 x 512 x 64 rows x 128 channels and
    128 x 128 x 128 rows x 256 channels are exactly 2 GiB).  Here EVERY row of such a tensor is live: the last row's last chunk is
    read / stored, and both arithmetic modes keep the bench-shape gates."""
    from tests import split_cases as sc
    case = TWO_GIB[kind](sc)
    bad = check_case(case, "two_gib." + kind)
    assert not bad, "\n".join(bad)
