import sys, struct, numpy as np
sys.path.insert(0,'/root/repo/tests')
import conftest
pkg = conftest.load_package()
c = pkg.Cascade('/root/repo/ugproj-detector_amd/data/cascades/haarcascade_frontalface_alt.hcc')
st, sn, tn, nr, nt, nl, nrr, lf = c.dump()
nr = nr.reshape(-1, 15)
def fhex(x): return "__uint_as_float(0x%08xu)" % struct.unpack('<I', struct.pack('<f', float(x)))[0]
def geom(ys, band_rows=16, win=20):
    cols = 63*ys + win + 1
    if ys == 1: return cols, 0
    half = (cols+1)//2
    return 2*half, half
def word(dx, dy, ys, pitch, half):
    col = dx if ys == 1 else ((dx & 1)*half + (dx >> 1))
    return dy*pitch + col
NST = int(sys.argv[1]) if len(sys.argv) > 1 else 6
out = []
out.append("// generated: specialised early stages of haarcascade_frontalface_alt")
out.append("#define HC_SPEC_NSTAGES %d" % NST)
for ys in (1, 2):
    pitch, half = geom(ys)
    node = 0
    for s in range(len(sn)):
        cnt = int(sn[s])
        if s < NST:
            out.append("template <> __device__ __forceinline__ bool spec_stage<%d, %d>(const int32_t* t, float vnf) {" % (ys, s))
            out.append("  double acc = 0.0;")
            for k in range(cnt):
                n = node + k
                terms = []
                for r in range(3):
                    x, y, w, h, wi = [int(v) for v in nr[n, 5*r:5*r+5]]
                    wt = float(wi)
                    if r == 2 and wt == 0.0: continue
                    o = [word(x, y, ys, pitch, half), word(x+w, y, ys, pitch, half), word(x, y+h, ys, pitch, half), word(x+w, y+h, ys, pitch, half)]
                    terms.append("%s * (float)(t[%d] - t[%d] - t[%d] + t[%d])" % (fhex(wt), o[0], o[1], o[2], o[3]))
                expr = "%s + %s" % (terms[0], terms[1])
                out.append("  { float v = %s;" % expr)
                if len(terms) == 3: out.append("    v += %s;" % terms[2])
                # stump: tree n has leaves lf[2n], lf[2n+1]
                out.append("    v *= vnf; acc += (double)(v < %s ? %s : %s); }" % (fhex(nt[n]), fhex(lf[2*n]), fhex(lf[2*n+1])))
                if k % 2 == 1: out.append("  __builtin_amdgcn_sched_barrier(0);")
            out.append("  return !(acc < (double)%s);" % fhex(st[s]))
            out.append("}")
        node += cnt
    out.append("template <> __device__ __forceinline__ bool spec_eval<%d>(int s, const int32_t* t, float vnf) {" % ys)
    out.append("  switch (s) {")
    for s in range(NST): out.append("    case %d: return spec_stage<%d, %d>(t, vnf);" % (s, ys, s))
    out.append("    default: return false;\n  }\n}")
open('/tmp/proto/spec_stages.h','w').write("\n".join(out)+"\n")
print("stumps specialised:", int(sn[:NST].sum()))
