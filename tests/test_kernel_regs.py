"""Build-time spill accounting (the +5.7 ms lesson of round 5: nine spilled registers in the weight-gradient GEMM that no
per-change A/B could see).  Every object of the in-tree build is unbundled, the AMDGPU metadata note of its gfx950 code object is read per kernel
(`scripts/kernel_regs.py`: llvm-objcopy -> clang-offload-bundler -> llvm-readelf --notes, llvm-objdump -d for WHERE the scratch instructions sit) and
the kernels a C2 / C4 step dispatches are held to: no VGPR spill, no scratch.  CPU only: hipcc cross-compiles here."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

# What a C2 train step launches (profiles/r05_trainstep_v5_kernel_stats.csv, every kernel above 0.02 % of the step + the STFT family) and
# the forward kernels of the f16 mode / inference.  Names as scripts/kernel_regs.py prints them (demangled, without the parameter list).
C2_STEP = [
    "gemm_tn_dual224_kernel<2, false>",
    "urse::lstm_bwd_nsplit_kernel<392, 0, 0>",
    "urse::lstm_bwd_kernel<unsigned short, 2, 4, 8, 0, 392, 0, 1, 1>",
    "urse::lstm_fwd_rwx_kernel<392, 416, 224, true, unsigned short, false>",      # the band path's forward beside a CU reservation (otherwise: the cluster forward in rounds)
    "urse::lstm_fwd_clusterx_kernel<unsigned short, false, true, 4>",
    "urse::lstm_fwd_clusterx_kernel<unsigned short, false, false, 4>",
    "urse::lstm_fwd_clusterx_kernel<unsigned short, false, false, 1>",      # small-batch inference: one / two row tiles per step (round 6)
    "urse::lstm_fwd_clusterx_kernel<unsigned short, false, false, 2>",
    "urse::lstm_fwd_clusterx_kernel<unsigned short, false, true, 1>",
    "urse::lstm_fwd_clusterx_kernel<unsigned short, false, true, 2>",
    "urse::lstm_fwd_clusterx_kernel<unsigned short, false, false, 3>",
    "urse::lstm_fwd_clusterx_kernel<unsigned short, false, true, 3>",
    "gemm_tn_dma_kernel<7, 2, 0>",
    "gemm_tn_dma_kernel<7, 2, 1>",          # f16-forward training: fc gradient against the f16 h (round 6)
    "gemm_nt_dma_gnb_kernel",
    "gemm_nt_dma_kernel<float, 7, 0, 256, 2, unsigned short>",
    "gemm_nt_bres_kernel<0, unsigned short>",
    "urse::gn_bwd_apply_kernel",
    "urse::gemm_tn_grouped_kernel<unsigned short>",
    "urse::gn_apply_kernel<unsigned short>",
    "gemm_nt_dma_grouped_kernel<unsigned short, 2, unsigned short>",
    "gemm_nt_dma_grouped_kernel<unsigned short, 1, unsigned short>",
    "gemm_nt_dma_grouped_kernel<float, 0, unsigned short>",
    "urse::gemm_nt_kernel<unsigned short, float>",
    "urse::gn_stats_kernel",
    "urse::gn_bwd_reduce4_kernel",
    "urse::clip_adamw_slots_kernel",
    "urse::bandsplit_apply_kernel<unsigned short>",
    "urse::bandsplit_stats_kernel",
    "urse::glu_mask_apply_bwd_kernel<unsigned short>",
    "urse::glu_mask_apply_kernel",
    "urse::mrl1_spec_reg_kernel<32, 24, true>",
    "urse::mrl1_spec_reg_kernel<32, 16, true>",
    "urse::mrl1_spec_reg_kernel<32, 32, true>",
    "urse::mrl1_spec_reg_kernel<16, 16, true>",
    "urse::stft960p_kernel<0, 2>",
    "urse::stft960p_kernel<1, 2>",
    "urse::istft960_kernel",
    "urse::lstm_pack_multi_kernel<unsigned short, unsigned short>",
]
# C4 (BSRNN-Flow, H = 768): the cluster forward and the other kernels of its step that C2 does not launch
C4_STEP = [
    "urse::lstm_fwd_cluster2_kernel<24, 8, 1, unsigned short, false>",      # training (saves): without the step's third barrier (round 6)
    "urse::lstm_fwd_cluster2_kernel<24, 8, 1, unsigned short, true>",       # forward only: with it
    "_ZN4urse24lstm_fwd_cluster2_kernelILi24ELi8ELi1EDF16_Lb1EEEvNS_12Cluster2ArgsE",      # the f16 forward-only instance (this c++filt does not know DF16_)
    "gemm_tn_dma_kernel<8, 2, 0>",
    "urse::lstm_bwd_split_kernel<1, 3>",      # the cooperative split BPTT at H = 768: 16 rows per cluster (urse_lstm_split_plan(768, 96) -> rows 16), three unit tiles per wave
]
# Spills that exist, where they sit, and the bound they are held to (a regression of any of them fails this test):
#  * (the fused cluster forward - all 256 registers, 160 of them resident weights - had ONE spilled register until round 6; in its rounds form, with the
#    helper wave's loop in front of the weight loads and the thread id opaque per round, it has none and sits in C2_STEP; its SGPR spills go to
#    lanes of a VGPR - v_writelane / v_readlane, no memory)
#  * the 32-ROW forms of the split BPTT (<2, *>: H <= 512, no benchmarked configuration dispatches them - C4's H = 768 runs <1, 3>, spill-free, in the
#    list above) spill ~50 registers inside their step loop on the 128-register budget of 16 waves: tracked so that it cannot grow unnoticed.
KNOWN = {
    "urse::lstm_bwd_split_kernel<2, 3>": dict(vgpr_spill_count=50, scratch_ops_in_mfma_loops=88),
    "urse::lstm_bwd_split_kernel<2, 2>": dict(vgpr_spill_count=52, scratch_ops_in_mfma_loops=90),
    #  * the mixed-operand dual weight-gradient GEMM (f16-forward training): two registers of its set-up are parked in scratch before the K loop
    "gemm_tn_dual224_kernel<2, true>": dict(vgpr_spill_count=2, scratch_ops_in_mfma_loops=0),
}


@pytest.fixture(scope="module")
def table(lib):
    import kernel_regs
    t = kernel_regs.collect()
    assert len(t) > 200, "the unbundler found %d kernels: tool chain broken?" % len(t)
    return t


def test_shipping_kernels_spill_nothing(table):
    missing = [k for k in C2_STEP + C4_STEP + list(KNOWN) if k not in table]
    assert not missing, ("kernels renamed / no longer instantiated - update the list", missing)
    bad = {k: {f: table[k][f] for f in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
           for k in C2_STEP + C4_STEP
           if table[k]["vgpr_spill_count"] != 0 or table[k]["private_segment_fixed_size"] != 0 or table[k]["scratch_ops"] != 0}
    assert not bad, bad


def test_known_spills_stay_where_they_are(table):
    for k, bound in KNOWN.items():
        for field, hi in bound.items():
            assert table[k][field] <= hi, (k, field, table[k][field], hi)


def test_register_table_is_committed_for_this_round(table):
    """profiles/r06_kernel_regs.json is regenerated by `python scripts/kernel_regs.py`; the committed copy must describe the same kernels (names) and
    the same spill counts as the library this test just built - a stale table is a table nobody looked at."""
    path = os.path.join(ROOT, "profiles", "r06_kernel_regs.json")
    assert os.path.exists(path), "run python scripts/kernel_regs.py"
    committed = json.load(open(path))["kernels"]
    assert set(committed) == set(table), sorted(set(committed) ^ set(table))[:10]
    drift = {k: (committed[k]["vgpr_spill_count"], table[k]["vgpr_spill_count"]) for k in table
             if committed[k]["vgpr_spill_count"] != table[k]["vgpr_spill_count"]}
    assert not drift, drift
