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
    if (argc > 8 && (atoi(argv[7]) == 4 || atoi(argv[7]) == 5)) {
        // MC-DPSK plug-in in gui::StreamingDecoder's order on ONE object (streaming_decoder.cpp:723,733,831,903-917,1347-1363):
        // configure / setSpreadingMode -> reset -> detectSync | detectDataSync(known CFO) -> setFrequencyOffset(sync CFO, with the
        // host's "trust the established CFO" rule) -> process(from start_sample) -> getSoftBits -> robustDecodeSingleCW.
        // argv: <modulation> <carriers> <buffer.f32> <known_cfo> <spreading> <out_prefix> 4 <data_sync>
        const int carriers = atoi(argv[2]), spreading = atoi(argv[5]);
        GpuMcDpskWaveform wf(carriers);
        wf.configure(mod, CodeRate::R1_4);
        wf.setSpreadingMode(spreading == 4 ? SpreadingMode::TIME_4X : spreading == 2 ? SpreadingMode::TIME_2X : SpreadingMode::NONE);
        std::string p = argv[6];
        if (atoi(argv[7]) == 5) {
            // HARQ chain of one data codeword (streaming_decoder.cpp:2758-2800): argv[8] receptions of equal length back to back in
            // the file, each training + reference + data: robust decode of the fresh soft bits; on failure ChaseCache::store
            // (first reception copies, later ones add) and, from the second reception on, robust decode of the sum
            const int n_rx = atoi(argv[8]);
            const size_t len = x.size() / static_cast<size_t>(n_rx);
            std::vector<float> acc; int combines = 0;
            for (int t = 0; t < n_rx; ++t) {
                wf.reset();
                wf.setFrequencyOffset(0.0f);
                if (!wf.process(SampleSpan{x.data() + t * len, len})) return 5;
                std::vector<float> soft = wf.getSoftBits();
                if (soft.size() < 648) return 6;
                soft.resize(648);
                int tries = 0, tries2 = 0;
                auto res = robustDecodeSingleCW(wf.handle(), soft.data(), &tries);
                if (!res.first) {
                    if (combines == 0) acc = soft; else for (size_t i = 0; i < 648; ++i) acc[i] += soft[i];   // chase_cache.cpp:27-88
                    ++combines;
                    if (combines > 1) res = robustDecodeSingleCW(wf.handle(), acc.data(), &tries2);
                }
                printf("%d %d %d %d %.9g", t + 1, res.first ? 1 : 0, tries, tries2, wf.getFadingIndex());
                if (res.first) for (int b = 0; b < 20; ++b) printf(" %u", res.second[b]);
                printf("\n");
                if (res.first) break;
            }
            return 0;
        }
        printf("%d %d %d %d\n", wf.getMinSamplesForFrame(), wf.getMinSamplesForCWCount(3), wf.getDataPreambleSamples(), wf.getPreambleSamples());
        {   // TX side of the plug-in: FNV-1a over the audio bytes of both preambles followed by modulate(bytes 0..80)
            Bytes coded(81);
            for (int i = 0; i < 81; ++i) coded[i] = static_cast<uint8_t>(i * 37 + 11);
            for (int which = 0; which < 2; ++which) {
                std::vector<float> a = which ? wf.generateDataPreamble() : wf.generatePreamble();
                std::vector<float> d = wf.modulate(coded);
                a.insert(a.end(), d.begin(), d.end());
                uint64_t hsh = 1469598103934665603ull;
                const unsigned char* b = reinterpret_cast<const unsigned char*>(a.data());
                for (size_t i = 0; i < a.size() * 4; ++i) { hsh ^= b[i]; hsh *= 1099511628211ull; }
                printf("%zu %llu\n", a.size(), static_cast<unsigned long long>(hsh));
            }
        }
        wf.reset();
        const float known = static_cast<float>(atof(argv[4]));
        const bool data_sync = atoi(argv[8]) != 0;
        SyncResult r;
        bool ok = data_sync ? wf.detectDataSync(SampleSpan{x.data(), x.size()}, r, known, 0.2f) : wf.detectSync(SampleSpan{x.data(), x.size()}, r, 0.15f);
        printf("%d %d %.9g %.9g\n", ok ? 1 : 0, r.start_sample, r.correlation, r.cfo_hz);
        if (!ok || r.start_sample < 0 || static_cast<size_t>(r.start_sample) >= x.size()) return 0;
        float new_cfo = r.cfo_hz;
        if (data_sync && std::abs(known) > 0.01f && std::abs(new_cfo - known) > 1.0f) new_cfo = known;
        wf.setFrequencyOffset(new_cfo);
        bool ready = wf.process(SampleSpan{x.data() + r.start_sample, x.size() - static_cast<size_t>(r.start_sample)});
        std::vector<float> soft = wf.getSoftBits();
        printf("%d %zu %.9g %.9g %.9g %d\n", ready ? 1 : 0, soft.size(), wf.estimatedCFO(), wf.getFadingIndex(), wf.getFrequencyOffset(), wf.isSynced() ? 1 : 0);
        FILE* o = fopen((p + ".llr").c_str(), "wb"); fwrite(soft.data(), 4, soft.size(), o); fclose(o);
        if (soft.size() >= 648) {
            int tries = 0;
            auto res = robustDecodeSingleCW(wf.handle(), soft.data(), &tries);
            printf("%d %d", res.first ? 1 : 0, tries);
            if (res.first) for (int b = 0; b < 20; ++b) printf(" %u", res.second[b]);
            printf("\n");
        }
        return 0;
    }
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
