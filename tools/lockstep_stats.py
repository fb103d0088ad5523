"""Development: per-iteration (failed sweeps, line-search halvings) of the oracle on the bench batch, and what lock-step execution of
several instances per wavefront would cost: run with NMPC_ORACLE_TRACE=1 captured per instance.
    python tools/lockstep_stats.py six 256
"""
import os, subprocess, sys, json, re
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(name, i0, i1):
    from tests import helpers as Hh
    from oracle import oracle_lib as O
    ocfg, B, P, W0 = Hh.bench_batch(name, i1)
    oc = O.make_config(ocfg)
    for b in range(i0, i1):
        sys.stderr.write("INST %d\n" % b); sys.stderr.flush()
        O.solve_batch(oc, P[b:b + 1], W0[b:b + 1], 1)

if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4])); sys.exit(0)
    name, n = sys.argv[1], int(sys.argv[2])
    env = dict(os.environ, NMPC_ORACLE_TRACE="1", OMP_NUM_THREADS="1")
    nproc = 8
    procs = [subprocess.Popen([sys.executable, __file__, "child", name, str(n * k // nproc), str(n * (k + 1) // nproc)], env=env, stderr=subprocess.PIPE, text=True) for k in range(nproc)]
    seqs = {}
    for p in procs:
        cur = None
        for ln in p.stderr:
            if ln.startswith("INST"):
                cur = int(ln.split()[1]); seqs[cur] = []
            elif ln.startswith("it "):
                m = re.search(r"alpha (\S+) a_p (\S+) .* ntry (\d+)", ln)
                al, ap, nt = float(m.group(1)), float(m.group(2)), int(m.group(3))
                nls = 1 + (int(round(np.log2(ap / al))) if al > 0 else 30)
                seqs[cur].append((nt + 1, nls))
        p.wait()
    its = np.array([len(seqs[b]) for b in range(n)])
    sw = np.array([sum(s for s, _ in seqs[b]) for b in range(n)]); ls = np.array([sum(l for _, l in seqs[b]) for b in range(n)])
    print("instances %d: mean iters %.2f max %d; sweeps/iter %.3f; merit evals/iter %.3f" % (n, its.mean(), its.max(), sw.sum() / its.sum(), ls.sum() / its.sum()))
    # lock-step cost model: an iteration of a wave costs  max_g sweeps * cs + max_g ls * cl + cr  (cs, cl, cr: sweep, merit, rest)
    for ipw in (1, 2, 4):
        for cs, cl, cr, tag in ((0.8, 0.04, 0.16, "m=6"), (0.45, 0.12, 0.43, "m=2")):
            # static pairing (instances b*ipw..): every group runs to the longest of its wave, no refill
            tot_static = 0.0; tot_ideal = 0.0
            for w0 in range(0, n - ipw + 1, ipw):
                grp = [seqs[b] for b in range(w0, w0 + ipw)]
                L = max(len(g) for g in grp)
                for t in range(L):
                    act = [g[t] for g in grp if t < len(g)]
                    tot_static += max(a[0] for a in act) * cs + max(a[1] for a in act) * cl + cr
                for g in grp:
                    tot_ideal += sum(a[0] * cs + a[1] * cl + cr for a in g)
            # refill: a queue feeds ipw slots of one wave; per step cost = max over active slots
            def refill(nw):
                q = list(range(n)); waves = [[None] * ipw for _ in range(nw)]; pos = [[0] * ipw for _ in range(nw)]; t_w = [0.0] * nw
                done = False
                # simulate each wave independently pulling from the shared queue in time order
                import heapq
                hp = [(0.0, w) for w in range(nw)]
                while hp:
                    t, w = heapq.heappop(hp)
                    for g in range(ipw):
                        if waves[w][g] is None and q:
                            waves[w][g] = q.pop(0); pos[w][g] = 0
                    act = [(g, seqs[waves[w][g]][pos[w][g]]) for g in range(ipw) if waves[w][g] is not None]
                    if not act:
                        t_w[w] = t; continue
                    c = max(a[0] for _, a in act) * cs + max(a[1] for _, a in act) * cl + cr
                    for g, _ in act:
                        pos[w][g] += 1
                        if pos[w][g] >= len(seqs[waves[w][g]]): waves[w][g] = None
                    heapq.heappush(hp, (t + c, w))
                return max(t_w), sum(t_w)
            nw = max(1, n // (ipw * 8))
            mk, busy = refill(nw)
            print("  ipw %d %s: static lock-step work x%.3f of ideal; refill (%d waves): busy x%.3f of ideal, makespan %.1f vs ideal/nw %.1f" %
                  (ipw, tag, tot_static / tot_ideal, nw, busy / tot_ideal, mk, tot_ideal / ipw / nw))
