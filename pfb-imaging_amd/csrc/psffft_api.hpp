// psffft_api.hpp -- PSF convolution on the hand-written row FFT (defined in psffft.hip).
//
// out = [out +] beam * crop( c2r( f(psfhat) * r2c( pad(beam * x) ) ) ) * scale + eta * x   as three row passes:
//   1. the nx non-zero rows of the padded image, two real rows per complex transform: forward FFT
//      along y, un-mixed through LDS, half spectrum kept                                        -> T1 (nx, nyo2)
//      transpose                                                                                -> T2 (nyo2, nx)
//   2. per y-frequency: zero-padded forward FFT along x, times f(psfhat) / N, inverse FFT along x in the
//      same registers, the first nx outputs kept                                                -> T2 (in place)
//      transpose back                                                                           -> T1
//   3. the nx output rows, two per transform (Z = A + i B): Hermitian-extended inverse FFT along y, real and
//      imaginary parts, crop, beam, scale, eta
// Nothing outside the nx x nyo2 corner is ever stored: ~6 GB of traffic at 8192^2 / 16384^2 where the
// padded r2c / c2r pipeline moves ~24 GB.  Padded sizes: ny_psf any plain row-FFT size ({1,3,5,7,9,15} x 2^a, 1024..16384);
// nx_psf a power of two (the two transforms of pass 2 chain in registers) or {3,5} x 2^a <= 10240 (they hand the row
// over through LDS).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "rowfft_api.hpp"

namespace pfbhip {

struct PsfFFT {
    bool ok = false;
    int64_t nx = 0, ny = 0, nxp = 0, nyp = 0, nyo2 = 0;
    size_t ld1 = 0;  // row stride of T1 (complex elements)
    size_t ld2 = 0;  // row stride of T2
    RowFFT fy, fx;
    DevBuf<double2> t1, t2;
    bool init(int64_t nx, int64_t ny, int64_t nxp, int64_t nyp);
    // psfhat (nxp, nyo2) real or complex on the device -> (nyo2, nxp), the layout pass 2 reads
    void transpose_psf(const double *src_dev, bool is_complex, double *dst_dev, hipStream_t st) const;
    void apply(const double *x_dev, const double *beam_dev, const double *psfT_dev, bool is_complex, int mode, double shift,
               double scale, double eta, int accumulate, double *out_dev, hipStream_t st);
    size_t device_bytes() const { return t1.bytes() + t2.bytes(); }
};

}  // namespace pfbhip
