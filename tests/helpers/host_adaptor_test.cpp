// C++ host-side use of the drop-in, written like tools/test_waveform_simple.cpp drives the reference:
// configure -> reset -> setAbsoluteTrainingPosition -> setFrequencyOffset -> process -> getSoftBits
// -> decodeFixedFrame (streaming_decoder.cpp:723,896,1347-1363,2821).
// usage: host_adaptor_test <mod> <rate> <frame.f32> <cfo_hz> <abs_pos> <out_prefix>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../ria_amd/host/gpu_waveform.hpp"
using namespace ria_host;
int main(int argc, char** argv) {
    if (argc < 7) return 2;
    auto mod = static_cast<Modulation>(atoi(argv[1]));
    auto rate = static_cast<CodeRate>(atoi(argv[2]));
    FILE* f = fopen(argv[3], "rb");
    if (!f) return 3;
    std::vector<float> x(1 << 20);
    x.resize(fread(x.data(), 4, x.size(), f));
    fclose(f);
    GpuOfdmChirpWaveform rx(mod, rate);
    GpuHandle dec(mod, rate);
    rx.reset();
    if (argc > 7 && atoi(argv[7]) == 2) {   // OFDM-COX: generatePreamble + detectSync twice on the same object (the
        GpuOfdmCoxWaveform cox(mod, rate);   // second call starts from the first one's noise floor, as the reference's does)
        printf("%zu\n", cox.generatePreamble().size());
        for (int pass = 0; pass < 2; ++pass) {
            SyncResult r;
            bool ok = cox.detectSync(SampleSpan{x.data(), x.size()}, r, static_cast<float>(atof(argv[4])));
            printf("%d %d %.9g %.9g\n", ok ? 1 : 0, r.start_sample, r.cfo_hz, r.correlation);
        }
        return 0;
    }
    if (argc > 8 && atoi(argv[7]) == 3) {
        // burst group in gui::StreamingDecoder's order on ONE waveform object (streaming_decoder.cpp:723-733, 896, 1347-1407,
        // 3127-3208): detectDataSync -> setAbsoluteTrainingPosition -> per frame setFrequencyOffset / process /
        // getSoftBits / estimatedCFO (2 Hz drift clamp); wasBurstInterleaved() is read AFTER the first process().
        // argv: <cfo_hz> <abs_base> <out_prefix> 3 <n_frames>
        const int n_frames = atoi(argv[8]);
        float cfo = static_cast<float>(atof(argv[4]));
        SyncResult r;
        bool ok = rx.detectDataSync(SampleSpan{x.data(), std::min<size_t>(x.size(), 21000)}, r, cfo, 0.5f);
        const int before = rx.wasBurstInterleaved() ? 1 : 0;
        printf("%d %d %.9g %d\n", ok ? 1 : 0, r.start_sample, r.correlation, before);
        if (!ok) return 0;
        const size_t frame_len = static_cast<size_t>(rx.getMinSamplesForFrame());
        size_t pos = static_cast<size_t>(r.start_sample);
        rx.setAbsoluteTrainingPosition(static_cast<size_t>(atoll(argv[5])) + pos);
        std::string p = argv[6];
        for (int f = 0; f < n_frames; ++f) {
            rx.setFrequencyOffset(cfo);
            if (pos + frame_len > x.size()) return 4;
            bool ready = rx.process(SampleSpan{x.data() + pos, frame_len});
            std::vector<float> soft = rx.getSoftBits();
            float corrected = rx.estimatedCFO();
            printf("%d %zu %.9g %.9g %d\n", ready ? 1 : 0, soft.size(), cfo, corrected, rx.wasBurstInterleaved() ? 1 : 0);
            float drift = corrected - cfo;
            if (std::abs(drift) > 2.0f) corrected = cfo + std::copysign(2.0f, drift);
            cfo = corrected;
            FILE* o = fopen((p + "." + std::to_string(f) + ".llr").c_str(), "wb"); fwrite(soft.data(), 4, soft.size(), o); fclose(o);
            pos += frame_len;
        }
        return 0;
    }
    if (argc > 7) {   // sync mode: detectDataSync / detectSync on the span, print the SyncResult, exit
        SyncResult r;
        bool ok = (atoi(argv[7]) == 1) ? rx.detectDataSync(SampleSpan{x.data(), x.size()}, r, static_cast<float>(atof(argv[4])), 0.5f)
                                       : rx.detectSync(SampleSpan{x.data(), x.size()}, r, 0.15f);
        printf("%d %d %.9g %.9g %d\n", ok ? 1 : 0, r.start_sample, r.correlation, r.cfo_hz, rx.wasBurstInterleaved() ? 1 : 0);
        return 0;
    }
    rx.setAbsoluteTrainingPosition(static_cast<size_t>(atoll(argv[5])));
    rx.setFrequencyOffset(static_cast<float>(atof(argv[4])));
    bool ready = rx.process(SampleSpan{x.data(), x.size()});
    float snr = rx.estimatedSNR(), cfo = rx.estimatedCFO(), fi = rx.getFadingIndex();
    std::vector<float> soft = rx.getSoftBits();
    CodewordStatus st = decodeFixedFrame(rx, dec, soft, true);
    std::string p = argv[6];
    FILE* o = fopen((p + ".llr").c_str(), "wb"); fwrite(soft.data(), 4, soft.size(), o); fclose(o);
    o = fopen((p + ".txt").c_str(), "w");
    fprintf(o, "%d %zu %.9g %.9g %.9g\n", ready ? 1 : 0, soft.size(), snr, cfo, fi);
    for (int cw = 0; cw < 4; ++cw) {
        fprintf(o, "%d %d", st.decoded[cw] ? 1 : 0, st.iterations[cw]);
        for (uint8_t b : st.data[cw]) fprintf(o, " %u", b);
        fprintf(o, "\n");
    }
    fclose(o);
    printf("ready=%d soft=%zu decoded=%d%d%d%d\n", ready, soft.size(), (int)st.decoded[0], (int)st.decoded[1], (int)st.decoded[2], (int)st.decoded[3]);
    return 0;
}
