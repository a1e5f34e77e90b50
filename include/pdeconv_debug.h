/* pdeconv_debug.h -- unit-test and measurement entry points of libpdeconv.so that are NOT part of the drop-in surface
 * (include/pdeconv.h): none of them stands in for a callable of the reference.  Used by tests/, bench.py and tools/ only;
 * the Julia glue (the .jl files under julia/) binds nothing from this file. */
#ifndef PDECONV_DEBUG_H
#define PDECONV_DEBUG_H

#include "pdeconv.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Unit-test entry of the register-resident wave FFT the fluid kernels are built on (csrc/wave_fft.hpp): nlines
 * lines of `len` complex doubles (device), natural order in and out, unnormalised forward (sgn < 0) / inverse
 * (sgn > 0) with FFTW's conventions (src/fluid_rk4.jl uses FFTW's fft / ifft).  len in {128,256,384,512,768}. */
int pdec_debug_wave_fft(const void* in_dev, void* out_dev, int len, int nlines, int sgn);

/* Measurement aid of bench.py (no reference counterpart): arm = 1 makes the NEXT fused critic pass launched on `critic`
 * (the behaviour critic of a 3-layer pair, src/PDEagent.jl:385-400) record s_memtime / s_memrealtime stamps at its phase
 * boundaries; arm = 0 copies the record of that launch to out13[13] (host): the mean shader cycles of the ten phases per
 * workgroup, their sum, the shader clock in GHz (d s_memtime / d s_memrealtime x 100 MHz) and the workgroup count.
 * Synchronises the stream of the pass. */
int pdec_debug_critic_stamps(pdec_handle critic, int arm, double* out13);
/* Measurement aid (no counterpart in the reference): the fp32 2-D Keller-Segel tile kernel on `nb` trajectories with `reps` RK4
 * sub-steps per launch on the tile held in registers (no halo refresh: timing only), `iters` launches between two events ->
 * microseconds per launch.  What a time-resident form of KellerSegelSetup.jl:213-239 x 32 could at best cost (HISTORY.md round 5). */
int pdec_debug_kseg2d_probe(pdec_handle env, int nb, int reps, int iters, double* us_per_launch);
/* (built only with -DPDEC_DEBUG_PROBES -- `make EXTRA=-DPDEC_DEBUG_PROBES OBJDIR=... OUT=...` --: the PROBE instantiation of the
 * tile kernel is not in the default library, where this entry returns PDEC_E_INVALID) */

/* Measurement aid of bench.py --emulate-ar-us (no reference counterpart): ONE workgroup of one wave that spins on the
 * constant-rate 100 MHz counter for `us` microseconds (<= 10 000) on `hip_stream` -- a stand-in for the latency of a
 * small-message collective on a box with one GPU.  The loop is bounded by the counter AND by an iteration cap. */
int pdec_debug_spin_us(void* hip_stream, double us);

#ifdef __cplusplus
}
#endif
#endif /* PDECONV_DEBUG_H */
