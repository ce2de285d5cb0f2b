"""CPU baseline leg of bench.py (TEST INFRASTRUCTURE: runs the oracle's tuned C/OpenMP restatement on the host cores).

Run as a child process so that the OpenMP runtime starts with its own environment (thread binding, active waiting) —
`python oracle/cpu_leg.py <grid side> <arnoldi steps> <matfree 0|1> <seconds budget>` prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ns, arnoldi, matfree, budget_s = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3])), float(sys.argv[4])
    import numpy as np
    from oracle import c_oracle as CO
    CO.build()
    # thread count: the runtime offers every hardware thread of the affinity mask, but SMT siblings and a container's CPU
    # quota (the GPU box hands a job far fewer CPUs than its affinity mask lists: beyond ≈ 32 threads the barriers wait on
    # descheduled threads) make "all of them" a bad default for a streaming code: take the count with which one step of THIS workload runs
    # fastest (and report the STREAM triad measured with the count that is best for the triad)
    avail = CO.num_threads()
    cands = sorted({max(1, avail // 16), max(1, avail // 8), max(1, avail // 4), max(1, avail // 2), avail})
    n = ns * ns
    nnz = 5 * n - 4 * ns
    z0 = np.zeros(n)
    use_csr = not matfree

    def run(alg, k):
        """k fixed-work Newton steps with one of the two Arnoldi processes the device has; returns the step loop's seconds"""
        if alg == "dcgs2":
            return CO.bratu_newton_fast(ns, 6.0, 0.0, z0, k, use_csr=use_csr, m=arnoldi)
        return CO.bratu_newton_fast_sstep(ns, 6.0, 0.0, z0, k, use_csr=use_csr, m=arnoldi, s=15, basis="newton")

    # ---- thread scan with SUSTAINED samples (≈ 0.4 s each, not two steps: a container's CPU quota lets a burst of all threads
    # run several times faster than the same threads can run for a second — round 2's two-step scan saw 194 steps/s where the
    # 9 s sample then ran 42)
    scan, triad = {}, 0.0
    best_t, best_rate = avail, 0.0
    per_scan = max(0.25, 0.03 * budget_s)
    for t in cands:
        CO.set_num_threads(t)
        triad = max(triad, CO.stream_triad(1 << 25, 2))
        t1 = run("dcgs2", 1)[2]                                   # placement / warm-up, and a first estimate
        k = int(max(3, min(400, per_scan / max(t1, 1e-4))))
        rate = k / run("dcgs2", k)[2]
        scan[t] = round(rate, 2)
        if rate > 1.03 * best_rate:
            best_t, best_rate = t, rate
    # … and the scan's winner is VALIDATED before it is used: a scan sample can still be a burst (seen in round 4: 247 steps/s in
    # the scan at 128 threads, 24 sustained). Candidates in the scan's order take one longer run each (≈ 8 % of the budget); the
    # first whose sustained rate keeps ≥ 70 % of its scan figure wins, else the one with the best sustained rate.
    validated = {}
    for t in sorted(scan, key=lambda q: -scan[q])[:4]:
        CO.set_num_threads(t)
        t1 = run("dcgs2", 1)[2]
        k = int(max(5, min(400, 0.08 * budget_s / max(t1, 1e-4))))
        validated[t] = k / run("dcgs2", k)[2]
        if validated[t] >= 0.7 * scan[t]:
            break
    best_t = max(validated, key=lambda q: validated[q])
    CO.set_num_threads(best_t)
    cores = best_t
    b_op = (12.0 * nnz + 4.0 * (n + 1) + 16.0 * n) if not matfree else 24.0 * n
    # bytes of a fixed-work step. delayed CGS2: at Arnoldi step k the dot sweep reads k+2 columns, the axpy sweep reads k+2 and
    # writes 2. s-step (blocks of 15 behind k = 1, 16 columns): three sweeps over k + 15 columns + two writes of 15 per block.
    # Once per Newton step: x = V y, Jacobian values, update, residual
    once = 8.0 * n * (arnoldi + 3) + 8.0 * nnz + 16.0 * n + 40.0 * n
    bytes_per_step = {
        "dcgs2": sum(b_op + 8.0 * n * (k + 2) + 8.0 * n * (k + 4) for k in range(arnoldi)) + once,
        "sstep": arnoldi * b_op + sum(8.0 * n * (3 * (k + 15) + 2 * 15) for k in (1, 16)) + once}
    triad = max(triad, CO.stream_triad(1 << 26, 3))
    spmv = CO.spmv_rate(ns, 8)
    # ---- the samples: for each Arnoldi process 5 runs of ≥ 50 steps (budget permitting) at the chosen thread count; the
    # median is the figure, min / max beside it
    algs = ["dcgs2", "sstep"] if arnoldi == 30 else ["dcgs2"]
    per_sample = 0.8 * budget_s / (5 * len(algs))
    samples, fn_last = {}, None
    for alg in algs:
        t1 = run(alg, 1)[2]
        k = int(max(5, min(400, per_sample / max(t1, 1e-4))))
        k = max(k, 50) if 50 * t1 <= 2.5 * per_sample else k
        rates = []
        for _ in range(5):
            _, fn, tk = run(alg, k)
            rates.append(k / tk)
            fn_last = float(fn[-1])
        rates.sort()
        samples[alg] = {"median": round(rates[2], 3), "min": round(rates[0], 3), "max": round(rates[-1], 3), "steps_per_sample": k,
                        "effective_GBs": round(rates[2] * bytes_per_step[alg] * 1e-9, 1)}
    best_alg = max(samples, key=lambda a: samples[a]["median"])
    rate = samples[best_alg]["median"]
    CO.set_num_threads(1)
    _, _, ts = run("dcgs2", 1)
    CO.set_num_threads(cores)
    eff = samples[best_alg]["effective_GBs"]
    print(json.dumps({
        "value": round(rate, 4), "unit": "newton_steps/s", "cores": cores, "kind": "port", "algorithm": best_alg,
        "sample": f"median of 5 runs of {samples[best_alg]['steps_per_sample']} fixed-work Newton steps of the same Bratu {ns}x{ns} workload "
                  f"({arnoldi} Arnoldi steps of GMRES each; the faster of the device's two Arnoldi processes restated in C: "
                  f"delayed CGS2 = oracle/nk_oracle.c::orc_bratu_newton_fast, s-step with the Newton basis = …_sstep2), OpenMP on "
                  f"{cores} of {avail} hardware threads (count chosen by sustained samples of the workload itself; "
                  f"OMP_PLACES={os.environ.get('OMP_PLACES')}, OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}, first-touch placement)",
        "samples": samples, "effective_GBs": eff, "stream_triad_GBs": round(triad, 1),
        "frac_of_stream_triad": round(eff / triad, 3) if triad > 0 else None,
        "spmv_GBs": round(spmv, 1), "spmv_frac_of_triad": round(spmv / triad, 3) if triad > 0 else None,
        "single_thread_value": round(1.0 / ts, 4), "thread_scan_steps_per_s": scan,
        "thread_count_validation_steps_per_s": {str(k_): round(v_, 2) for k_, v_ in validated.items()},
        "thread_scan_best": max(scan.values()), "sample_over_scan_best": round(samples["dcgs2"]["median"] / max(scan.values()), 3),
        "sample_over_scan_at_chosen_count": round(samples["dcgs2"]["median"] / scan[cores], 3),
        "fnorm_inf_last": fn_last,
        "note": "restatement of the reference algorithm (Julia is not installed on this box); a reported baseline, not the target"}))


if __name__ == "__main__":
    main()
