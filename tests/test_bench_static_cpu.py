"""-m "not gpu": the committed files bench.py's static fields come from still describe the kernels bench.py names.  `roofline.traffic`
(fabric bytes) and `roofline.issue` (instruction counts, instruction mix) cannot be measured inside the bench process -- counters need
rocprofv3 around it -- so they are file constants under profiles/; a renamed or removed kernel must fail HERE instead of silently dropping
out of the sums (pmc_traffic / issue_bound skip names they do not find)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _kernels_of(csv_path):
    names = set()
    for line in open(csv_path):
        if line.startswith("#") or line.startswith("kernel,"):
            continue
        names.add(line.split(",")[0])
    return names


def test_every_stage_kernel_is_in_the_committed_counter_summaries():
    import bench
    for csv_name in (bench.PMC_TRAFFIC_CSV, bench.PMC_VALU_CSV):
        path = bench._pmc_path(csv_name)
        assert path is not None, f"no committed counter summary for {csv_name} under profiles/"
        have = _kernels_of(path)
        for stage, kernels in bench.PMC_STAGE_KERNELS.items():
            for k in kernels:
                assert k in have, f"bench.PMC_STAGE_KERNELS[{stage!r}] names {k}, which {os.path.basename(path)} does not hold: renamed kernel?"


def test_stage_kernels_exist_in_the_sources():
    """... and the names are kernels of the current sources (a summary of an older round may hold kernels that are gone)."""
    import bench
    import re
    src = "".join(open(os.path.join(ROOT, "emd_amd", "csrc", f)).read() for f in ("preprocess.hip", "binning.hip", "render.hip"))
    for stage, kernels in bench.PMC_STAGE_KERNELS.items():
        for k in kernels:
            base = re.sub(r"\[[ND]\]$", "", k)
            assert re.search(r"\b" + re.escape(base) + r"\s*\(", src), f"{k} ({stage}) is not a kernel of emd_amd/csrc"


def test_issue_mix_file_covers_the_render_kernels():
    """bench.issue_bound reads the static instruction mix of K6 / K7 from profiles/r06_render_isa_mix.txt (made by profiles/make_isa_mix.py from the
    disassembly of the shipped library), not from literals."""
    import bench
    mix = bench.load_isa_mix()
    for k in ("k_render_backward_q", "k_render_forward_q"):
        assert k in mix and 0.0 <= mix[k]["wide_fraction"] <= 1.0 and mix[k]["valu"] > 50, (k, mix.get(k))


def test_committed_isa_mix_is_the_mix_of_the_current_source():
    """profiles/r06_render_isa_mix.txt is what profiles/make_isa_mix.py prints for the render.hip in the tree (hipcc -S cross-compiles gfx950 on CPU):
    the static instruction mix bench.py prices K6 / K7 with cannot go stale behind a kernel change."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "make_isa_mix.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    pick = lambda text: [l for l in text.splitlines() if l.startswith(("MIX ", "LOOP "))]
    committed = pick(open(os.path.join(ROOT, "profiles", "r06_render_isa_mix.txt")).read())
    assert committed and pick(out.stdout) == committed


def test_fine_stage_kernel_models_price_every_hip_kernel_of_the_step():
    """profiles/fine_stage.py: every k_* kernel name the fine-stage step launches (names as torch.profiler reports them) matches a model, MFMA kernels
    are priced in FLOPs against the split-bf16 peak, HBM kernels in bytes against 8 TB/s; torch's own kernels are reported without a bound."""
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import fine_stage as fs
    M = fs.kernel_models(2_000_000, 1_300_000, 3_900_000, 1066 * 1600)
    names = ["void (anonymous namespace)::k_hexplane_bwd_agg<32, 2>(EmdHexArgs, EmdHexGrads, unsigned int)",
             "void (anonymous namespace)::k_hexplane_bwd_plane<32>(EmdHexArgs, EmdHexGrads)", "void (anonymous namespace)::k_hexplane_fwd4<32>(EmdHexArgs, unsigned int)",
             "void (anonymous namespace)::k_mlp_branch_bwd<1, 2, true, false, 1, true>(EmdMlpBranch, EmdMlpBranchGrads)",
             "void (anonymous namespace)::k_mlp_branch_bwd<1, 1, false, true, 0, false>(EmdMlpBranch, EmdMlpBranchGrads)",
             "void (anonymous namespace)::k_mlp_branch_fwd<1, 1, false, false>(EmdMlpBranch)", "void (anonymous namespace)::k_mlp_branch_fwd<1, 2, true, true>(EmdMlpBranch)",
             "void (anonymous namespace)::k_mlp_trunk_fwd<4>(EmdMlpTrunk)", "void (anonymous namespace)::k_mlp_trunk_bwd<4, false>(EmdMlpTrunk, EmdMlpTrunkGrads)",
             "void (anonymous namespace)::k_mlp_embed_bwd<4, false>(EmdMlpTrunk, EmdMlpTrunkGrads)", "void (anonymous namespace)::k_preprocess<0, true>(PreArgs)",
             "(anonymous namespace)::k_preprocess_backward(PreBwdArgs)", "void (anonymous namespace)::k_render_forward_q<true, 0, false>((anonymous namespace)::RenderDims, unsigned int const*)",
             "void (anonymous namespace)::k_render_backward_q<false, false, 0, false>((anonymous namespace)::RenderDims)", "(anonymous namespace)::k_ssim_forward(EmdLossArgs)",
             "(anonymous namespace)::k_ssim_backward(EmdLossArgs)", "(anonymous namespace)::k_loss_pointwise(EmdLossArgs, float*)", "(anonymous namespace)::k_sky_backward(EmdSkyBwdArgs)"]
    for n in names:
        e = fs.price(n, 0.5, M)
        assert e["bound"] in ("hbm", "mfma"), n
        assert 0 < e["frac"] < 1.0, (n, e)
        assert ("TFLOPs" in e) == ("k_mlp_trunk" in n or "k_mlp_branch" in n), n
    e = fs.price("void at::native::vectorized_elementwise_kernel<4, at::native::CUDAFunctor_add<float>, std::array<char*, 3ul> >", 0.007, M)
    assert e["bound"] is None and "frac" not in e
    assert fs.short_name(names[0]) == "k_hexplane_bwd_agg<32, 2>"
