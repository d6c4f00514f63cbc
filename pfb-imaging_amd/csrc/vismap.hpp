// vismap.hpp -- the per-visibility position map: THE bit-exact contract shared with the
// CPU oracle (oracle/pfb_oracle.c: pfbo_vismap).  Every statement is one IEEE-754 double
// operation; this header must be compiled with -ffp-contract=off (no fused multiply-add),
// and so is the oracle.
//
//   fc      = freq[c] / c0                       (host, double division)
//   (u,v,w) = uvw * (su,sv,sw) * fc
//   flip    = do_w && w < 0  ->  (u,v,w) = -(u,v,w)           [Hermitian fold, w >= 0]
//   xu      = u * pixsize_x ;  fu = xu - floor(xu) ;  pu = fu * nu     (grid coordinate in [0, nu])
//   iu0     = (int) floor(pu + (1 - W/2))        first of W taps (may be < 0; taps wrap mod nu)
//   pw      = (w - wmin) * (1/dw) ;  p0 = (int) floor(pw + (1 - W/2))
//   tile    = (wrap(iu0, nu) / T) * ntv + wrap(iv0, nv) / T
// The gridder runs the TRANSPOSED problem internally (swap_uv: its "u" is the caller's v); every formula
// above is symmetric in the two axes, so the caller's (iu0, iv0) are the plan's (iv0, iu0), bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pfbhip {

constexpr int TILE = 32;  // uv tile edge in grid cells

struct MapArgs {
    const double *uvw;  // (nrow,3)
    const double *fc;   // (nchan) freq / c
    const uint8_t *mask;  // (nrow,nchan) or nullptr
    int64_t nvis;       // nrow * nchan
    int nchan;
    double su, sv, sw;
    double px, py;
    double dnu, dnv;  // (double) nu, nv
    int nu, nv;
    int ntv;          // tiles along v
    int key_planes;   // 1: sort key = tile; P > 1: key = tile * P + first plane (wide fields, ES-kernel planes)
    int key_sub;      // 64: the key carries the 4 x 4-cell block of the footprint origin inside its tile in its low 6 bits
                      // (run order of the register-footprint scatter, k_grid_blk); 1: no sub-key
    double shift;     // 1 - W/2
    int W;
    int do_w;
    double wmin, xdw;
    int swap_uv;      // 1: the plan works on the transposed problem (u <-> v exchanged at the C-ABI boundary, see gridder.hip)
};

struct VisPos {
    double u, v, w;     // wavelengths, after flips and the Hermitian fold
    double pu, pv, pw;  // grid coordinates
    int iu0, iv0, p0;
    int flip;
};

__device__ __forceinline__ int wrap_index(int i, int n)
{
    i %= n;
    return i < 0 ? i + n : i;
}

__device__ __forceinline__ VisPos vis_position(const MapArgs &m, int64_t i)
{
    VisPos r;
    int64_t row = i / m.nchan;
    int chan = int(i - row * m.nchan);
    double f = m.fc[chan];
    const int cu = m.swap_uv ? 1 : 0;
    double u = m.uvw[3 * row + cu] * m.su;
    double v = m.uvw[3 * row + 1 - cu] * m.sv;
    double w = m.uvw[3 * row + 2] * m.sw;
    u = u * f;
    v = v * f;
    w = w * f;
    r.flip = 0;
    if (m.do_w && w < 0.0) {
        u = -u;
        v = -v;
        w = -w;
        r.flip = 1;
    }
    r.u = u;
    r.v = v;
    r.w = w;
    double xu = u * m.px;
    double xv = v * m.py;
    double fu = xu - floor(xu);
    double fv = xv - floor(xv);
    r.pu = fu * m.dnu;
    r.pv = fv * m.dnv;
    r.iu0 = (int)floor(r.pu + m.shift);
    r.iv0 = (int)floor(r.pv + m.shift);
    if (m.do_w) {
        double t = w - m.wmin;
        r.pw = t * m.xdw;
        r.p0 = (int)floor(r.pw + m.shift);
    } else {
        r.pw = 0.0;
        r.p0 = 0;
    }
    return r;
}

__device__ __forceinline__ uint32_t tile_of(const MapArgs &m, int iu0, int iv0)
{
    return uint32_t(wrap_index(iu0, m.nu) / TILE) * uint32_t(m.ntv) + uint32_t(wrap_index(iv0, m.nv) / TILE);
}

}  // namespace pfbhip
