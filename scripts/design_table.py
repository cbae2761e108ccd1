#!/usr/bin/env python3
"""DESIGN.md section 8 (SURVEY 8 row by row) with its "measured" column taken from the newest bench line on file.

    python scripts/design_table.py            rewrite the block between the design_table markers of DESIGN.md
    python scripts/design_table.py --check    exit 1 if DESIGN.md does not hold what this script would write

Sources, newest round first: profiles/rNN_bench_line.json (the full JSON line of `python bench.py` on one MI355X, copied
from gpurun_out/ by hand when a round's numbers are final) and profiles/rNN_bench_line_{2,6}ranks_one_gpu.json (the
multi-rank rehearsals).  Only tracked files under profiles/ are read: the driver's BENCH_rNN.json lands after the last
commit of a round and must not be able to change what --check expects.  Everything else in a row (where it is built, what tests it) is static
text below: a row is edited HERE, never in DESIGN.md, which is why the table cannot go stale behind a bench line.
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- design_table:begin -->", "<!-- design_table:end -->"


def newest(pattern):
    best = None
    for path in glob.glob(os.path.join(ROOT, pattern)):
        m = re.search(r"r(\d+)", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    return best


def load_line(path):
    text = open(path).read()
    try:
        return json.loads(text)
    except ValueError:      # a captured stdout: the line is the one that starts with {
        return json.loads([l for l in text.split("\n") if l.startswith("{")][0])


def sci(x, digits=3):
    """1.907e7 -> '1.91·10⁷'"""
    if x is None:
        return "—"
    mant, exp = ("%.*e" % (digits - 1, x)).split("e")
    sup = str(int(exp)).translate(str.maketrans("-0123456789", "⁻⁰¹²³⁴⁵⁶⁷⁸⁹"))
    return "%s·10%s" % (mant, sup)


def rows():
    rnd, path = newest("profiles/r*_bench_line.json")
    d = load_line(path)
    src = os.path.relpath(path, ROOT)
    roof, cpu = d["roofline"], d["cpu_baseline"]
    oc = d["other_configs"]
    chain = {k.split(" ")[0] + (" null" if "null" in k else ""): v for k, v in oc.items()}
    wf, share = d["workflow_config3"], d["workflow_config3_share_of_8"]
    mid = d["mid_batch_half_step"]
    c4 = chain["configs[4]"]
    multi = []
    for n in (2, 6):
        got = newest("profiles/r*_bench_line_%dranks_one_gpu.json" % n)
        if got:
            m = load_line(got[1])
            inv = m.get("workflow_config3_sharded", {}).get("world_size_invariance", {})
            multi.append("%d ranks on one card (`%s`): whole command %.0f s, sharded workflow p = %.6f, in-run invariance check %s"
                         % (n, os.path.relpath(got[1], ROOT), m["_rehearsal"]["wall_clock_of_the_whole_command_s"],
                            m["workflow_config3_sharded"]["p_value"],
                            "identical" if inv.get("identical_to_one_rank_alone") else "NOT RUN"))
    head = ("`%s`: **%s evals/s** at N = 10⁴, J = 6 (%.2f ms per 512 000 rows; kernel `%s` %.2f ms); roofline %.3f of "
            "algorithmic HBM bytes, %.3f of the FP64 vector peak%s; HBM traffic %s; worst difference to the CPU port %.1e over %d "
            "rows of the timed batch; CPU port %s evals/s on %d cores, %s on one; null model (J = 3) %s; three SHO terms (six "
            "ranks of arithmetic) %s, FP64 fraction %.2f; 32 000 rows: %.2f / %.2f ms on the pipeline against %.2f / %.2f "
            "one-lane; PCIe-inclusive %s"
            % (src, sci(d["value"]), d["ms_per_step"], roof["kernel"], roof["kernel_ms"], roof["frac"],
               roof["fp64_valu"]["frac"],
               (", **%.2f of its FP64 issue floor at the clock held** (%d MHz)" % (roof["fp64_issue_frac_at_clock"], roof["fp64_issue"]["sclk_mhz"]))
               if roof.get("fp64_issue_frac_at_clock") else "",
               "%.2f GB per launch = %.1f %% of algorithmic" % (roof["traffic"] / 1e9, 100 * roof["traffic"] / (roof["evals_per_launch"] * roof["bytes_per_eval"]))
               if roof.get("traffic") else "see `profiles/r04_pmc_traffic.json`",
               cpu["max_rel_diff_vs_gpu"], cpu["compared"], sci(cpu["value"]), cpu["cores"], sci(cpu["single_thread"]),
               sci(d["null_model_sweep"]["evals_per_s"]), sci(d["true_J6_sweep"]["evals_per_s"]), d["true_J6_sweep"]["fp64_valu_frac"],
               mid["alt_J5_arith"]["pipeline_ms"], mid["null_J3"]["pipeline_ms"], mid["alt_J5_arith"]["one_lane_ms"],
               mid["null_J3"]["one_lane_ms"], sci(d["end_to_end"]["value"])))
    its = "configs[0] / [1] / [2] / [2] null / [4]: %s / %s / %s / %s / %.0f iterations/s" % (
        sci(chain["configs[0]"]["iterations_per_s"]), sci(chain["configs[1]"]["iterations_per_s"]),
        sci(chain["configs[2]"]["iterations_per_s"]), sci(chain["configs[2] null"]["iterations_per_s"]), c4["iterations_per_s"])
    ws = c4["walker_shard_8"]
    proj = share.get("projected_speedup_at_8_gpus")
    share_s = share.get("whole_test_s_with_observed_split", share["whole_test_s"])
    table = [
        ("a1–a8 log-probability (GP set-up, prior, coefficients, factorisation + solve)",
         "`csrc/mtg_sweep.h`, `mtg_kernels.hip`, `mtg_kernels_multi.hip`, `mtg_sweep_pipe.h`, `mtg_kernels_pipe.hip`, `mtg_prepare.h`, "
         "`mtg_sort.hip`, `mtg_timeparallel*`, `mtg_tp_scan.*`, `mtg_tp_big*`, `mtg_capi.hip`; `gp.py`, `gpmodelling.py`",
         "`test_hip_parity`, `test_golden_gpu` (180 dense/mpmath vectors × 3 dispatch modes), `test_box_golden`, `test_fuzz_gpu`, "
         "`test_edge_cases_gpu`, `test_timeparallel_gpu`, `test_tp_big_gpu`, `test_window_gpu`, `test_sort_gpu`, `test_device_math_gpu`, "
         "`test_highfreq_golden_gpu`, `test_pipe_gpu` (pipeline and paired launch bit-identical to the one-lane sweep; 32 000 rows vs "
         "the oracle); CPU: `test_oracle`, `test_modeling_terms`, `test_capi_cpu`", head),
        ("a9 `fit`", "`GPModelling.fit`, `ppp.batched_minimize`",
         "`test_gpmodelling_gpu::test_fit_improves_and_matches_oracle_at_optimum`, `test_ppp_cpu`",
         "one launch of P + 1 rows per L-BFGS-B iteration; line search in 3 launches"),
        ("a10 `derive_posteriors`", "`sampler.py`, `device_sampler.py`, `csrc/mtg_sampler.hip`",
         "`test_sampler` (emcee semantics + the independent draw-order replay), `test_device_sampler_gpu` + `philox_replay` (device chain "
         "replayed on the host with the oracle; shipped speculative shapes; resume bit for bit), `test_gpmodelling_gpu`",
         its + "; T_LRT of configs[2] = %.3f" % chain["configs[2]"]["T_LRT"]["value"]),
        ("a11 `spread_walkers`, a12 accessors", "`walkers.py`, `gpmodelling.py`",
         "`test_spread_golden` (**the reference's own seeded outputs**, 54 arrays, identity), `test_gpmodelling_cpu` (the reference's "
         "`tests/gpmodelling_test.py` restated)", "—"),
        ("b boundary", "`include/mtg.h`, `libmtg_hip.so`, `engine.py`, `examples/c_api_demo.c`, `INTEGRATION.md`",
         "`test_capi_cpu` (exports = header, fail-loud), `test_capi_example` (plain C99 vs the oracle on the GPU)",
         "host-pointer entry point %s evals/s (%.0f %% of the resident rate)" % (sci(d["end_to_end"]["value"]), 100 * d["end_to_end"]["value"] / d["value"])),
        ("c oracle", "`oracle/dense.py`, `oracle/celerite_ref.c`, `tests/golden/`",
         "`test_oracle`, `test_psd_models`, `test_stats_io`, `test_spread_golden`, `test_coeff_golden` (reference outputs for PSDs, "
         "information criteria, spread_walkers, coefficient builders)", "**parity unpinned at the lnL boundary** for J > 0 values (celerite absent); what the notebooks print from celerite is reproduced (`test_notebook_known_answer`: a white-kernel lnL to 1.3·10⁻⁸; celerite's posterior maxima on two rebuilt light curves within 10⁻⁴ / 2·10⁻² of this build's maximum), §2"),
        ("d measurement", "`bench.py`, `scripts/profile_bench.sh`, `scripts/summarize_profile.py`, `scripts/design_table.py`, `profiles/`",
         "bench aborts on non-finite values or > 10⁻⁸ difference to the port; `test_bench_cpu` (launcher, this table in sync)",
         "measured HBM copy %.2f TB/s (headline = %.2f of it); clock under load %s MHz, %s W"
         % (d["hbm_copy_measured"]["GB_per_s"] / 1e3, roof.get("frac_of_copy_measured", float("nan")),
            d.get("clock_under_load", {}).get("sclk_mhz", "?"), d.get("clock_under_load", {}).get("socket_power_w", "?"))),
        ("e multi-GPU", "`distributed.py`, `ppp.protassov_test(sharded=True)`, `derive_posteriors(shard_walkers=True)` "
         "(`mtg_ensemble_shard_rccl` / `_host`), `bench.py --gpus N`",
         "`test_distributed` (gloo world 2 and 3: every sharding bit-equal to one process, both splits of the Protassov test, failure "
         "propagation; RCCL with one rank; two-GPU RCCL chain, skipped on one GPU)",
         "**N > 1 on hardware: not run here.**  One-GPU projections: sweep `strong_shard_8.per_gpu_factor` %.3f (× 8 = %.1f×); workflow "
         "share of 8: %.2f s of %.2f s ⇒ %.2f×; configs[4] walker shard 256 → 32 rows: %.2f → %.2f ms = %.2f×.  Rehearsals: %s"
         % (d["strong_shard_8"]["per_gpu_factor"], 8 * d["strong_shard_8"]["per_gpu_factor"], share_s, wf["whole_test_s"],
            proj if proj else wf["whole_test_s"] / share_s, ws["half_step_ms_256_rows"], ws["half_step_ms_32_rows"], ws["speedup"],
            "; ".join(multi) if multi else "none on file")),
        ("f1 device sampler + convergence", "`csrc/mtg_sampler.hip`, `device_sampler.py`, `sampler.integrated_time`, `mtg_chain_autocorr`",
         "as a10; `test_chain_autocorr_on_the_device_matches_the_host`",
         "one solve + one sampler launch per iteration (speculative) or per half-step, no host round trip"),
        ("f2 simulator", "`csrc/mtg_simulate.hip`, `simulator.py`, `models/psd_models.py`",
         "`test_simulator_gpu` (exact host replay, chirp-z vs library transform, moments, noise; `stream=numpy`: the reference notebook's own light curves for its seeds; "
         "the E13 flux-PDF adjustment on the device = the numpy loop to 10⁻¹² from the same white series; Kraft noise on the device = `add_noise` epoch by epoch), "
         "`test_simulator_reference_cases` (the reference's known answers, incl. its `test_pdf_lognormal / _uniform` at their own 10⁶ points through the device adjustment), `test_psd_models`, `test_ppp_gpu` (block invariance; a lognormal Protassov test that never enters the host loop)",
         "%d × %s-point simulations in %.2f s inside the workflow" % (wf["nsims"], sci(wf["fft_points_per_simulation"]), wf["seconds"]["simulate"])),
        ("f3 predict", "`mtg_predict_kernel`, `mtg_apply_inverse_kernel`, `GP.predict`, `standarized_residuals`",
         "`test_gpmodelling_gpu` (dense algebra)", "O(N·J²) instead of celerite's dense N × N"),
        ("f4 LRT / IO", "`stats.py`, `lightcurves.py`, `ppp.protassov_test`",
         "`test_stats_io` (bit for bit vs reference `stats.py` outputs), `test_ppp_gpu`, `test_distributed`",
         "configs[3] as a workflow: **%.1f s** (observed chains %.2f, simulation %.2f, refits %.1f + %.1f), p = %.6f, %s refit "
         "evaluations/s end to end" % (wf["whole_test_s"], wf["seconds"]["observed_chains"], wf["seconds"]["simulate"],
                                       wf["seconds"]["refit_null"], wf["seconds"]["refit_alt"], wf["p_value"],
                                       sci(wf["refit_evaluations_per_s_end_to_end"]))),
    ]
    return table


def render():
    out = ["| Row | Built in | Parity / behaviour tests | Measured (one MI355X unless noted) |", "|---|---|---|---|"]
    for row in rows():
        out.append("| " + " | ".join(cell.replace("|", "\\|") for cell in row) + " |")
    return "\n".join(out)


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    a, b = text.index(BEGIN) + len(BEGIN), text.index(END)
    new = text[:a] + "\n" + render() + "\n" + text[b:]
    if "--check" in sys.argv:
        if new != text:
            sys.stderr.write("DESIGN.md section 8 is stale: run python scripts/design_table.py\n")
            raise SystemExit(1)
        return
    open(path, "w").write(new)


if __name__ == "__main__":
    main()
