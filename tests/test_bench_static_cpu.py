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
