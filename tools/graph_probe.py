"""gogp_observe_gradient_candidates with the launch sequence on streams (graph = 0), as an explicitly built hipGraph
that is one chain in enqueue order (graph = 1, N <= 1024) and as one with the sweep's real dependencies as edges
(graph = 2, N <= 8192): evaluations per second, nodes of the graph, whether the results are bit-identical to the streams'.
usage: python3 tools/graph_probe.py [N,N,...] [k,k,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import gp as G, kernel, synth
Ns = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "512,1024,2048,4096,8192").split(",")]
ks = [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "1,8").split(",")]
D = 4
for N in Ns:
    X, y = synth.make_inputs(N, D, 20251114 + 1)
    g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
    base = np.log([1.0, np.sqrt(D / 6.0), 0.1])
    for k in ks:
        if k > 1 and N > 4096:
            continue
        ref = None
        for mode in (0, 1, 2, 3):
            if mode == 1 and N > 1024:
                continue
            g.set_option("graph", mode)
            xs = lambda r: np.array([base + 0.01 * ((r * k + c) % 7) for c in range(k)])
            for r in range(3):  # the graph is built on the second identical use
                g.observe_gradient_candidates(xs(r))
            reps = max(3, min(30, int(20000 / N)))
            t = time.perf_counter()
            for r in range(reps):
                lmls, grads, st = g.observe_gradient_candidates(xs(r))
            t = (time.perf_counter() - t) / reps
            lmls, grads, st = g.observe_gradient_candidates(xs(0))
            if ref is None:
                ref = (lmls.copy(), grads.copy())
            same = np.array_equal(ref[0], lmls) and np.array_equal(ref[1], grads)
            nodes, refused = g.graph_info()
            print("N %5d k %2d graph %d: %8.3f ms per launch sequence, %8.1f evals/s, nodes %4d%s, bit-identical to streams: %s" % (
                N, k, mode, t * 1e3, k / t, nodes, " (REFUSED)" if refused else "", same), flush=True)
    g.close()
