/* probe_sor.c -- CPU what-if model of the SOR kernel's number formats (development tool, not
 * shipped, not the oracle).  Lexicographic sweep (identical dependency states to the GPU's
 * hyperplane pipeline) with switchable storage/arithmetic precision so numerics can be explored
 * without GPU time.  Interior-only arrays (Z,Y,X), Neumann ghost = own old value. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define I3(z,y,x) (((size_t)(z)*Y+(y))*X+(x))

static double rf(double v, int on) { return on ? (double)(float)v : v; }

/* flags: bit0 J stored fp32; bit1 d stored fp32; bit2 fp32 arithmetic; bit3 psi from factors A (12
 * arrays, stored fp32 if bit0); bit4 L stored fp32; bit5 wpsi stored fp32; bit6: sweep J from factors */
/* switch_it: from this iteration on the storage roundings are off (operands re-derived in full precision, increments
 * and w psi stored in fp64): "the last iterations of a level in a wider format" (round 4 what-if).  >= iters: never. */
void sor_probe_sw(const double *const J[10], const double *const A[12], const double *weight,
                  const double *const Lin[3], int Z, int Y, int X, double ax, double ay, double az,
                  int iters, int lag, double adc, int flags, int switch_it, double *const dout[3])
{
    const int J32 = flags & 1, AR32 = (flags >> 2) & 1, PF = (flags >> 3) & 1,
              L32 = (flags >> 4) & 1, JF = (flags >> 6) & 1;
    int D32 = (flags >> 1) & 1, W32 = (flags >> 5) & 1;
    size_t n = (size_t)Z * Y * X;
    double *du = calloc(n, 8), *dv = calloc(n, 8), *dw = calloc(n, 8), *wpsi = malloc(n * 8);
    double *Jr[10], *Ar[12], *L[3];
    for (int a = 0; a < 10; a++) { Jr[a] = malloc(n * 8); for (size_t q = 0; q < n; q++) Jr[a][q] = rf(J[a][q], J32); }
    for (int a = 0; a < 12; a++) { Ar[a] = malloc(n * 8); for (size_t q = 0; q < n; q++) Ar[a][q] = rf(A[a][q], J32); }
    for (int a = 0; a < 3; a++) { L[a] = malloc(n * 8); for (size_t q = 0; q < n; q++) L[a][q] = rf(Lin[a][q], L32); }
    if (JF) { /* rebuild J from the (rounded) factors in double: J_ab = sum_k A_ka A_kb */
        /* order J11,J22,J33,J44,J12,J13,J23,J14,J24,J34 ; A[k*4+a] */
        static const int pa[10] = {0,1,2,3,0,0,1,0,1,2}, pb[10] = {0,1,2,3,1,2,2,3,3,3};
        for (int e = 0; e < 10; e++) for (size_t q = 0; q < n; q++) {
            double s = 0; for (int k = 0; k < 3; k++) s += Ar[k*4+pa[e]][q] * Ar[k*4+pb[e]][q];
            Jr[e][q] = s;
        }
    }
    const double OM = 1.95;
    for (int it = 0; it < iters; it++) {
        int upd = (it % lag) == 0;
        if (it == switch_it) {  /* wider storage from here on */
            D32 = W32 = 0;
            for (int a = 0; a < 10; a++) memcpy(Jr[a], J[a], n * 8);
            for (int a = 0; a < 12; a++) memcpy(Ar[a], A[a], n * 8);
            for (int a = 0; a < 3; a++) memcpy(L[a], Lin[a], n * 8);
            if (!upd) for (size_t q = 0; q < n; q++) wpsi[q] = wpsi[q];  /* the frozen weights stay as stored */
        }
        for (int k = 0; k < Z; k++) for (int j = 0; j < Y; j++) for (int i = 0; i < X; i++) {
            size_t c = I3(k, j, i);
            double u0 = du[c], v0 = dv[c], w0 = dw[c];
            if (upd) {
                double val;
                if (PF) {
                    val = 0;
                    for (int q = 0; q < 3; q++) {
                        double r = Ar[q*4+0][c]*u0 + Ar[q*4+1][c]*v0 + Ar[q*4+2][c]*w0 + Ar[q*4+3][c];
                        val += r * r;
                    }
                } else {
                    val = Jr[0][c]*u0*u0 + Jr[1][c]*v0*v0 + Jr[2][c]*w0*w0 + 2.0*Jr[4][c]*u0*v0 + 2.0*Jr[5][c]*u0*w0
                        + 2.0*Jr[6][c]*v0*w0 + 2.0*Jr[7][c]*u0 + 2.0*Jr[8][c]*v0 + 2.0*Jr[9][c]*w0 + Jr[3][c];
                }
                if (val < 0) val = 0;
                double ps = (adc != 1.0) ? adc * pow(val + 1e-6, adc - 1.0) : 1.0;
                wpsi[c] = rf(weight[c] * ps, W32);
            }
            double wps = wpsi[c];
#define NB(arr, cond, off, self) ((cond) ? arr[off] : (self))
            double sux = NB(du, i>0, c-1, u0) + NB(du, i<X-1, c+1, u0);
            double svx = NB(dv, i>0, c-1, v0) + NB(dv, i<X-1, c+1, v0);
            double swx = NB(dw, i>0, c-1, w0) + NB(dw, i<X-1, c+1, w0);
            double suy = NB(du, j>0, c-X, u0) + NB(du, j<Y-1, c+X, u0);
            double svy = NB(dv, j>0, c-X, v0) + NB(dv, j<Y-1, c+X, v0);
            double swy = NB(dw, j>0, c-X, w0) + NB(dw, j<Y-1, c+X, w0);
            size_t P = (size_t)Y * X;
            double suz = NB(du, k>0, c-P, u0) + NB(du, k<Z-1, c+P, u0);
            double svz = NB(dv, k>0, c-P, v0) + NB(dv, k<Z-1, c+P, v0);
            double swz = NB(dw, k>0, c-P, w0) + NB(dw, k<Z-1, c+P, w0);
            double diag = 2*ax + 2*ay + 2*az;
            if (AR32) {
                float fax=(float)ax, fay=(float)ay, faz=(float)az, w=(float)wps;
                float nu = fmaf(faz,(float)suz,fmaf(fay,(float)suy,fmaf(fax,(float)sux,(float)L[0][c])));
                float nv = fmaf(faz,(float)svz,fmaf(fay,(float)svy,fmaf(fax,(float)svx,(float)L[1][c])));
                float nw = fmaf(faz,(float)swz,fmaf(fay,(float)swy,fmaf(fax,(float)swx,(float)L[2][c])));
                float J11=(float)Jr[0][c],J22=(float)Jr[1][c],J33=(float)Jr[2][c],J12=(float)Jr[4][c],J13=(float)Jr[5][c],
                      J23=(float)Jr[6][c],J14=(float)Jr[7][c],J24=(float)Jr[8][c],J34=(float)Jr[9][c];
                float deu=fmaf(w,J11,(float)diag), dev=fmaf(w,J22,(float)diag), dew=fmaf(w,J33,(float)diag);
                float bu=w*J14, bv=w*J24, bw=w*J34;
                float fu0=(float)u0, fv0=(float)v0, fw0=(float)w0;
                float n2 = nu - bu; n2 -= w*fmaf(J13,fw0,J12*fv0);
                float u1 = fmaf(1.95f, n2/deu, (float)(1.0-OM)*fu0);
                n2 = nv - bv; n2 -= w*fmaf(J23,fw0,J12*u1);
                float v1 = fmaf(1.95f, n2/dev, (float)(1.0-OM)*fv0);
                n2 = nw - bw; n2 -= w*fmaf(J23,v1,J13*u1);
                float w1 = fmaf(1.95f, n2/dew, (float)(1.0-OM)*fw0);
                du[c]=u1; dv[c]=v1; dw[c]=w1;
            } else {
                double nu = L[0][c] + ax*sux + ay*suy + az*suz;
                double nv = L[1][c] + ax*svx + ay*svy + az*svz;
                double nw = L[2][c] + ax*swx + ay*swy + az*swz;
                double deu = diag + wps*Jr[0][c], dev = diag + wps*Jr[1][c], dew = diag + wps*Jr[2][c];
                double u1 = (1-OM)*u0 + OM*(nu - wps*(Jr[7][c] + Jr[4][c]*v0 + Jr[5][c]*w0))/deu;
                double v1 = (1-OM)*v0 + OM*(nv - wps*(Jr[8][c] + Jr[4][c]*u1 + Jr[6][c]*w0))/dev;
                double w1 = (1-OM)*w0 + OM*(nw - wps*(Jr[9][c] + Jr[5][c]*u1 + Jr[6][c]*v1))/dew;
                du[c]=rf(u1,D32); dv[c]=rf(v1,D32); dw[c]=rf(w1,D32);
            }
        }
    }
    memcpy(dout[0], du, n*8); memcpy(dout[1], dv, n*8); memcpy(dout[2], dw, n*8);
    for (int a = 0; a < 10; a++) free(Jr[a]);
    for (int a = 0; a < 12; a++) free(Ar[a]);
    for (int a = 0; a < 3; a++) free(L[a]);
    free(du); free(dv); free(dw); free(wpsi);
}

void sor_probe(const double *const J[10], const double *const A[12], const double *weight,
               const double *const Lin[3], int Z, int Y, int X, double ax, double ay, double az,
               int iters, int lag, double adc, int flags, double *const dout[3])
{
    sor_probe_sw(J, A, weight, Lin, Z, Y, X, ax, ay, az, iters, lag, adc, flags, iters + 1, dout);
}

/* ---- defect-correction ("delta") form of the same sweep (round 4) -------------------------------------------
 * d = d0 + delta.  At a psi-update iteration a voxel folds d0 <- d0 + delta (d0 in `d0fmt` precision), builds M
 * from the factors and stores the RESIDUAL of the window's linear system at d0
 *     r = L + sum_nb a (d0_nb - d0) - b4 - M d0
 * (-1 neighbours are already folded in this sweep, +1 neighbours are taken as d0 + delta, which is what they fold
 * to), and every iteration runs the reference's recurrence (level_solver_3d.py:503-540) on delta with r in place
 * of b.  M, r, delta may be held in fp32: their rounding is relative to |delta| and |r|, which shrink as the outer
 * iteration converges, not to |d| and |b|.
 * fmt: 0 = fp64, 1 = fp32, 2 = packed 42 bit (31 significant bits).  */
static double rq(double v, int fmt)
{
    if (fmt == 1) return (double)(float)v;
    if (fmt == 2) {
        unsigned long long b;
        memcpy(&b, &v, 8);
        b = (b + (1ull << 21)) & ~((1ull << 22) - 1ull);
        memcpy(&v, &b, 8);
        return v;
    }
    return v;
}
void sor_probe_dc(const double *const A[12], const double *weight, const double *const Lin[3], int Z, int Y, int X,
                  double ax, double ay, double az, int iters, int lag, double adc, int fmtM, int fmtR, int fmtD,
                  int fmtD0, int fmtA, int fmtL, double *const dout[3])
{
    size_t n = (size_t)Z * Y * X, P = (size_t)Y * X;
    double *d0[3], *dl[3], *M[6], *r[3], *Ar[12], *L[3];
    for (int a = 0; a < 3; a++) { d0[a] = calloc(n, 8); dl[a] = calloc(n, 8); r[a] = calloc(n, 8); }
    for (int a = 0; a < 6; a++) M[a] = calloc(n, 8);
    for (int a = 0; a < 12; a++) { Ar[a] = malloc(n * 8); for (size_t q = 0; q < n; q++) Ar[a][q] = rq(A[a][q], fmtA); }
    for (int a = 0; a < 3; a++) { L[a] = malloc(n * 8); for (size_t q = 0; q < n; q++) L[a][q] = rq(Lin[a][q], fmtL); }
    const double OM = 1.95, diag = 2 * ax + 2 * ay + 2 * az;
    for (int it = 0; it < iters; it++) {
        int upd = (it % lag) == 0;
        for (int k = 0; k < Z; k++) for (int j = 0; j < Y; j++) for (int i = 0; i < X; i++) {
            size_t c = I3(k, j, i);
            /* neighbour offsets; a ghost is the voxel itself (old value) */
            size_t nb[6] = {i > 0 ? c - 1 : c, i < X - 1 ? c + 1 : c, j > 0 ? c - X : c, j < Y - 1 ? c + X : c,
                            k > 0 ? c - P : c, k < Z - 1 ? c + P : c};
            const double aw[6] = {ax, ax, ay, ay, az, az};
            double dn[3][6]; /* delta seen from the six neighbours */
            double own[3] = {dl[0][c], dl[1][c], dl[2][c]};
            if (upd) {
                /* fold: the voxel's full increment becomes d0 (what does not fit d0's format stays in delta) */
                double full[3], nd0[3];
                for (int a = 0; a < 3; a++) { full[a] = d0[a][c] + dl[a][c]; nd0[a] = rq(full[a], fmtD0); own[a] = full[a] - nd0[a]; }
                double val = 0;
                for (int q = 0; q < 3; q++) {
                    double rr = Ar[q*4+0][c]*full[0] + Ar[q*4+1][c]*full[1] + Ar[q*4+2][c]*full[2] + Ar[q*4+3][c];
                    val += rr * rr;
                }
                double w = weight[c] * ((adc != 1.0) ? adc * pow(val + 1e-6, adc - 1.0) : 1.0);
                double m[6] = {0, 0, 0, 0, 0, 0}, b4[3] = {0, 0, 0};
                static const int pa[6] = {0, 1, 2, 0, 0, 1}, pb[6] = {0, 1, 2, 1, 2, 2};
                for (int q = 0; q < 3; q++) {
                    for (int e = 0; e < 6; e++) m[e] += w * Ar[q*4+pa[e]][c] * Ar[q*4+pb[e]][c];
                    for (int a = 0; a < 3; a++) b4[a] += w * Ar[q*4+a][c] * Ar[q*4+3][c];
                }
                /* d0 of the neighbours as they will be (or are) after THEIR fold of this sweep */
                double lap[3] = {0, 0, 0};
                for (int q = 0; q < 6; q++) {
                    int done = (q % 2 == 0) && nb[q] != c; /* -1 neighbours: already folded and relaxed */
                    for (int a = 0; a < 3; a++) {
                        double D0, Dl;
                        if (nb[q] == c) { D0 = nd0[a]; Dl = own[a]; }
                        else if (done) { D0 = d0[a][nb[q]]; Dl = dl[a][nb[q]]; }
                        else { double f = d0[a][nb[q]] + dl[a][nb[q]]; D0 = rq(f, fmtD0); Dl = f - D0; }
                        lap[a] += aw[q] * (D0 - nd0[a]);
                        dn[a][q] = Dl;
                    }
                }
                double rr[3];
                rr[0] = (L[0][c] - b4[0]) + lap[0] - (m[0]*nd0[0] + m[3]*nd0[1] + m[4]*nd0[2]);
                rr[1] = (L[1][c] - b4[1]) + lap[1] - (m[3]*nd0[0] + m[1]*nd0[1] + m[5]*nd0[2]);
                rr[2] = (L[2][c] - b4[2]) + lap[2] - (m[4]*nd0[0] + m[5]*nd0[1] + m[2]*nd0[2]);
                for (int e = 0; e < 6; e++) M[e][c] = rq(m[e], fmtM);
                for (int a = 0; a < 3; a++) { r[a][c] = rq(rr[a], fmtR); d0[a][c] = nd0[a]; }
            } else {
                for (int q = 0; q < 6; q++) for (int a = 0; a < 3; a++) dn[a][q] = nb[q] == c ? own[a] : dl[a][nb[q]];
            }
            double s[3];
            for (int a = 0; a < 3; a++) {
                s[a] = r[a][c];
                for (int q = 0; q < 6; q++) s[a] += aw[q] * dn[a][q];
            }
            double u1 = (1 - OM) * own[0] + OM * (s[0] - M[3][c] * own[1] - M[4][c] * own[2]) / (diag + M[0][c]);
            double v1 = (1 - OM) * own[1] + OM * (s[1] - M[3][c] * u1 - M[5][c] * own[2]) / (diag + M[1][c]);
            double w1 = (1 - OM) * own[2] + OM * (s[2] - M[4][c] * u1 - M[5][c] * v1) / (diag + M[2][c]);
            dl[0][c] = rq(u1, fmtD); dl[1][c] = rq(v1, fmtD); dl[2][c] = rq(w1, fmtD);
        }
    }
    for (int a = 0; a < 3; a++) for (size_t q = 0; q < n; q++) dout[a][q] = d0[a][q] + dl[a][q];
    for (int a = 0; a < 3; a++) { free(d0[a]); free(dl[a]); free(r[a]); free(L[a]); }
    for (int a = 0; a < 6; a++) free(M[a]);
    for (int a = 0; a < 12; a++) free(Ar[a]);
}
