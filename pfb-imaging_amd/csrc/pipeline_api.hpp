// pipeline_api.hpp -- internal (C++) entry points that let one driver chain the device-resident operators
// of different handles on ONE stream (used by the on-device primal-dual loop, pd.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct pfbhip_psi;
struct pfbhip_psfconv;

namespace pfbhip {

// Psi (psi.hip): all launches of the handle go to `st` until swapped back; returns the previous stream
hipStream_t psi_swap_stream(pfbhip_psi *p, hipStream_t st);
void psi_geometry(const pfbhip_psi *p, int64_t *nx, int64_t *ny, int *nbasis, int64_t *nxmax, int64_t *nymax);
void psi_dot_async(pfbhip_psi *p, const double *x_dev, double *alpha_dev);
void psi_hdot_async(pfbhip_psi *p, const double *alpha_dev, double *x_dev);

// l21 dual update phases and positivity (psi.hip), arrays (nband, n)
void l21_vtilde_async(const double *vp_dev, double *v_dev, int64_t nband, int64_t n, double sigma, double *sum_dev, hipStream_t st);
void l21_scale_async(double *v_dev, int64_t nband, int64_t n, double lam, const double *weight_dev, const double *sum_dev,
                     hipStream_t st);
// a = Psi^H xp in; a = updated dual, ext = 2 a - vp out (all bands on this device)
void l21_fused_async(const double *vp_dev, double *a_dev, double *ext_dev, int64_t nband, int64_t n, double lam, double sigma,
                     const double *weight_dev, hipStream_t st);
// bands spread over ranks: local band sum of vtilde, then (given the all-reduced sum) the update + extrapolation
void l21_localsum_async(const double *vp_dev, const double *a_dev, int64_t nband, int64_t n, double sigma, double *sum_dev,
                        hipStream_t st);
void l21_apply_async(const double *vp_dev, double *a_dev, double *ext_dev, int64_t nband, int64_t n, double lam, double sigma,
                     const double *weight_dev, const double *sum_dev, hipStream_t st);
void positivity_async(double *x_dev, int64_t nband, int64_t n, int mode, hipStream_t st);
// positivity mode 2 across ranks: flag pixels with a non-positive LOCAL band; zero all local bands where the (summed) flag > 0
void positivity_flag_async(const double *x_dev, int64_t nband, int64_t n, double *bad_dev, hipStream_t st);
void positivity_zero_async(double *x_dev, int64_t nband, int64_t n, const double *bad_dev, hipStream_t st);

// PSF convolution (psfconv.hip)
hipStream_t psfconv_stream(pfbhip_psfconv *p);
// all launches of the plan go to `st` until swapped back; returns the previous stream
hipStream_t psfconv_swap_stream(pfbhip_psfconv *p, hipStream_t st);
void psfconv_geometry(const pfbhip_psfconv *p, int64_t *nx, int64_t *ny);
void psfconv_apply_async(pfbhip_psfconv *p, const double *x_dev, int64_t psf_slot, int64_t beam_slot, int mode, double shift,
                         double scale, double eta, int accumulate, double *out_dev);

}  // namespace pfbhip
