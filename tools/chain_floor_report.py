"""profiles/round5_chain_floor.json (round-4 verdict, next-1a): the measured costs of the primitives of the Riccati sweeps on gfx950
(tools/chain_floor.hip -> profiles/round5_chain_floor_raw.json), and from them the dependent-chain length and the issue time of ONE stage step of
riccati_factor_rows (csrc/tmpc_riccati.hpp), set against the cycles the real kernel takes per stage (tmpc_debug_profile: profiles/round4_final_phases.jsonl
for the square-root form of rounds 1-4, profiles/round5_final_phases.jsonl for the input-block form of round 5).  Instruction mixes are those of the bench
kernel's factor loop (tools/isa_loops.py on tmpc_solve_compact_kernel<8,8,3>).
Usage: python tools/chain_floor_report.py > profiles/round5_chain_floor.json"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(os.path.join(ROOT, "profiles", "round5_chain_floor_raw.json")))
T = {(t["test"], t["chains_per_wave"], t["waves_per_simd"]): t for t in raw["tests"]}
def c(test, K=1, W=1, key="cycles_per_rep"):
    return T[(test, K, W)][key]

fma_dep = c("fma_f64")                                  # one dependent v_fma_f64
fma_issue_1w = c("fma_f64", 8) / 8                      # independent FMAs, one wave
fma_issue_2w = c("fma_f64", 8, 2) / 8 / 2               # per instruction per SIMD with two waves
dpp_dep = c("dpp_bcast_only"); dpp_issue = c("dpp_bcast_only", 4) / 4
rsq_dep = c("rsq_f64_only"); rsq_issue = c("rsq_f64_only", 4) / 4
col_1 = c("chol_column (bcast, rsqrt_nr, mul, bcast, fma)"); col_2 = c("chol_column (bcast, rsqrt_nr, mul, bcast, fma)", 2) / 2
col_2w = c("chol_column (bcast, rsqrt_nr, mul, bcast, fma)", 1, 2)

def phases(path, mode=0):
    for line in open(os.path.join(ROOT, "profiles", path)):
        d = json.loads(line)
        if d["latency_mode"] == mode:
            return d
    return None
p4 = phases("round4_final_phases.jsonl"); p5 = phases("round5_final_phases.jsonl")
N = 20
def per_stage(p, key):
    return p["cycles"][key] / p["mean_ipm_total"] / N

# one stage step of the factorisation: instruction mix of the loop body (VALU total, f64 arithmetic, DPP moves, v_rsq_f64, other VALU)
mix4 = dict(valu=227, f64=152, dpp=43, rsq=7, salu=39, lds=21)
mix5 = dict(valu=138, f64=74, dpp=28, rsq=2, salu=34, lds=21)
def issue_cycles(m):
    other = m["valu"] - m["f64"] - m["dpp"]
    return (m["f64"] - m["rsq"]) * fma_issue_1w + m["rsq"] * rsq_issue + m["dpp"] * dpp_issue + other * 4.5
def chain_cycles(pivots):
    # per pivot: the measured column pattern (broadcast pivot, rsqrt_nr, scale, broadcast, rank-1 update of the next pivot); between stages: broadcast of
    # the carried 5x5 block (one DPP move), the own-column product (5 dependent FMAs + the fma that joins the extra row), the row assembly (<= 5 dependent FMAs)
    return pivots * col_1 + dpp_dep + 6 * fma_dep + 5 * fma_dep

out = {
    "what": "measured floor of the Riccati chain on gfx950 (MI355X): primitive costs, critical path and issue time of one stage step of the factorisation vs the kernel's measured cycles",
    "sources": {"microbenchmark": "profiles/round5_chain_floor_raw.json (tools/chain_floor.hip, one workgroup on one CU, s_memtime of the slowest wave; 2.4 GHz)",
                "kernel_phase_clocks": ["profiles/round4_final_phases.jsonl", "profiles/round5_final_phases.jsonl"], "instruction_mix": "tools/isa_loops.py on tmpc_solve_compact_kernel<8,8,3,false,64,0>"},
    "primitive_costs_cycles": {
        "v_fma_f64 dependent (one wave)": fma_dep, "v_fma_f64 independent, per instruction (one wave, 8 chains)": fma_issue_1w,
        "v_fma_f64 per instruction per SIMD with two waves": fma_issue_2w, "v_mul_f64 dependent": c("mul_f64"), "v_add_f64 dependent": c("add_f64"),
        "v_mov_b64_dpp row_newbcast dependent (incl. the 2 wait states the hazard needs)": dpp_dep, "v_mov_b64_dpp independent, per instruction": dpp_issue,
        "v_mov_b64_dpp -> v_fma_f64 pair, dependent": c("dpp_bcast+fma_f64"), "v_readlane_b32 x 2 -> v_fma_f64, dependent (the round-2 cross-lane path)": c("readlane_pair+fma_f64"),
        "v_rsq_f64 dependent": rsq_dep, "v_rsq_f64 independent, per instruction (a quarter-rate instruction)": rsq_issue,
        "rsqrt_nr = v_rsq_f64 + Halley step (6 dependent operations)": c("rsqrt_nr (rsq + 5 ops)"),
        "one Cholesky column (pivot broadcast, rsqrt_nr, scale, column broadcast, update of the next pivot: 10 dependent operations)": col_1,
        "same, two independent columns interleaved in one wave, per column": col_2, "same, one chain per wave, two waves per SIMD, per column per wave": col_2w,
        "LDS ds_write_b64 -> ds_read_b64 -> use": c("lds write->read round trip + fma"), "LDS ds_add_f64 -> ds_read_b64 -> use": c("lds ds_add_f64 -> read + fma"),
        "LDS ds_read_b32, address from the previous read (load-to-use latency)": c("lds read, address from the value read")},
    "reading": ("A DEPENDENT f64 operation costs a wave %.2f cycles, an independent one %.2f: on this part the FP64 pipe itself is the chain -- a lone wave's dependent "
                "stream already runs at %.0f %% of the rate it could issue independent work at, and two waves per SIMD bring the SIMD to %.2f cycles per instruction "
                "(the nominal 4.0 is not reached by a pure FMA stream either).  So the sweeps were never waiting on latency that more parallelism could hide: they are "
                "ISSUE-bound at two waves per SIMD, and only fewer (or cheaper: v_rsq_f64 issues for %.1f cycles) instructions move them.") % (fma_dep, fma_issue_1w, 100 * fma_issue_1w / fma_dep, fma_issue_2w, rsq_issue),
    "factor_stage": {
        "square_root_form_rounds_1_to_4": {
            "instruction_mix": mix4, "pivots_on_the_chain": 7, "critical_path_cycles": chain_cycles(7), "issue_cycles_one_wave": issue_cycles(mix4),
            "measured_cycles_per_stage_one_wave_per_SIMD": per_stage(p4, "riccati_factor"),
            "note": "the measured 1.85 k cycles are 3 x the dependent chain and 1.45 x the VALU issue time of the loop (the rest: 39 SALU instructions, LDS operand waits, s_nop hazards): "
                    "issue, not the chain, is what a stage costs"},
        "input_block_form_round_5": {
            "instruction_mix": mix5, "pivots_on_the_chain": 2, "critical_path_cycles": chain_cycles(2), "issue_cycles_one_wave": issue_cycles(mix5),
            "measured_cycles_per_stage_one_wave_per_SIMD": per_stage(p5, "riccati_factor") if p5 else None}},
    "vector_sweeps": {
        "measured_cycles_per_stage_step_one_wave_per_SIMD": {"round4": per_stage(p4, "riccati_solve") / 3, "round5": (per_stage(p5, "riccati_solve") / 3) if p5 else None},
        "instructions_per_stage_step": {"forward": {"valu": 33, "lds": 12}, "backward": {"valu": 30, "lds": 9}},
        "note": "three sweeps per interior-point iteration (forward of the predictor, backward + forward of the corrector).  ~33 VALU instructions take ~350 cycles: here the bound is the "
                "operand traffic -- 9-12 LDS instructions per stage step whose results are needed within the same or the next step (LDS load-to-use ~70 cycles, in-order return)"},
    "whole_solve": None,
}
if p5:
    out["whole_solve"] = {"cycles_per_solve_one_wave_per_SIMD": {"round4": p4["cycles"]["total"], "round5": p5["cycles"]["total"]},
                          "share_of_the_sequential_phases": {"round4": (p4["cycles"]["riccati_factor"] + p4["cycles"]["riccati_solve"]) / p4["cycles"]["total"],
                                                             "round5": (p5["cycles"]["riccati_factor"] + p5["cycles"]["riccati_solve"]) / p5["cycles"]["total"]}}
print(json.dumps(out, indent=1))
