"""CPU study behind DESIGN section 6: |delta psi| of the device contract (oracle counter mode = the GPU bit
for bit) against the real reference, next to the reference against itself under another seed, on the
bench events.  Needs oracle/_ref (this container).  python tools/dpsi_study.py"""
import sys, os, numpy as np, multiprocessing as mp
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
def work(rng):
    devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1)
    from _libs import RefLib, OrcLib
    from miso_amd import workload
    R = RefLib(); O = OrcLib()
    out=[]
    for e in rng:
        exons, isoforms, pos, cig = workload.event_reads(e, 2, 1000, 36)
        fl=[c for ex in exons for c in ex]
        g = R.gene(fl, isoforms); og = O.gene(fl, isoforms)
        R.rng_seed(1000+e)
        r = R.miso(g, pos, cig, 36, iters=7500, burn=2500, lag=1, chains=1)
        R.rng_seed(5000+e)
        r2 = R.miso(g, pos, cig, 36, iters=7500, burn=2500, lag=1, chains=1)
        c = O.miso(og, pos, cig, 36, iters=7500, burn=2500, lag=1, chains=1, mode=OrcLib.COUNTER, seed=42, event_id=e)
        out.append((e, r.samples[:,0].mean(), r.samples[:,0].std(ddof=1), r2.samples[:,0].mean(), c.samples[:,0].mean(), c.samples[:,0].std(ddof=1)))
    return out
if __name__=="__main__":
    N=2400
    with mp.get_context("fork").Pool(8) as p:
        res=[x for part in p.map(work, [range(i, N, 8) for i in range(8)]) for x in part]
    a=np.array(res)
    sd=np.maximum(a[:,2],1e-12)
    z_gc=np.abs(a[:,4]-a[:,1])/sd; z_rr=np.abs(a[:,3]-a[:,1])/sd
    print("counter-vs-ref: mean|d| %.5f max %.5f max z %.2f | ref-vs-ref: mean|d| %.5f max %.5f max z %.2f" % (np.abs(a[:,4]-a[:,1]).mean(), np.abs(a[:,4]-a[:,1]).max(), z_gc.max(), np.abs(a[:,3]-a[:,1]).mean(), np.abs(a[:,3]-a[:,1]).max(), z_rr.max()))
    for i in np.argsort(-z_gc)[:6]:
        print("event %d ref %.4f (sd %.4f) ref2 %.4f counter %.4f (sd %.4f) z_gc %.2f z_rr %.2f" % (a[i,0],a[i,1],a[i,2],a[i,3],a[i,4],a[i,5],z_gc[i],z_rr[i]))
    print("signed mean diff counter-ref: %.6f +- %.6f ; ref2-ref: %.6f" % ((a[:,4]-a[:,1]).mean(), (a[:,4]-a[:,1]).std()/np.sqrt(len(a)), (a[:,3]-a[:,1]).mean()))
