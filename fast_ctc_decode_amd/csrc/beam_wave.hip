// beam_wave.hip -- register-resident CTC prefix beam search (beam_size <= 8, N <= 7).
// Placeholder until the kernel lands: reports "unsupported" so AUTO falls back to the
// LDS-resident kernel in beam_generic.hip.
#include "fcd_internal.h"

namespace fcd {

bool beam_wave_supported(int, int, int) { return false; }

hipError_t launch_beam_wave(const BatchDesc &, int64_t, int64_t, const BeamArgs &, const WaveArena &,
                            const ResultDesc &, hipStream_t) {
    return hipErrorNotSupported;
}

}  // namespace fcd
