// oracle/ref_shim_clisim.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The reference's simulator applies a transmitter frequency offset with SimulatedChannel::applyTxCFO
// (tools/cli_simulator.cpp:298-341: FFT of the next power of two -> frequency-domain Hilbert -> inverse FFT ->
// rotation by a wrapped float phase, real part).  SURVEY.md 8d names it as the CFO impairment of config 4.  It is a
// private member of a class defined inside the tool's own translation unit, so that file is compiled here where it
// lies (-I$(REF)), unmodified: its main() is renamed and made a static unused function, which the optimiser drops
// together with everything only main() reaches; what remains referenced is ultra::FFT (src/dsp/fft.cpp, already part of
// oracle/Makefile's REF_SRCS).  Nothing is copied from the reference; this file only calls it.
//
// The standard headers come first because '#define private public' must not be active while libstdc++ is parsed.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <complex>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <queue>
#include <random>
#include <set>
#include <span>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#define private public
#define main static __attribute__((unused)) ria_unused_cli_simulator_main
#include "tools/cli_simulator.cpp"
#undef main
#undef private

// phase_inout: SimulatedChannel::cfo_phase_a_to_b_ before / after the call (phase continuity across transmissions)
extern "C" int ref_apply_tx_cfo(const float* in, int n, float cfo_hz, float* phase_inout, float* out) {
    SimulatedChannel ch;
    ch.tx_cfo_hz_ = cfo_hz;
    std::vector<float> x(in, in + n);
    float phase = phase_inout ? *phase_inout : 0.0f;
    std::vector<float> y = ch.applyTxCFO(x, phase);
    if (phase_inout) *phase_inout = phase;
    std::memcpy(out, y.data(), static_cast<size_t>(n) * sizeof(float));
    return n;
}
