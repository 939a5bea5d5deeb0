// tests/host_fft_harness.cpp -- TEST HARNESS ONLY (built by tests/test_host_fft.py with the host clang++ into a temporary directory; never part of,
// linked into or loaded by the library).
// Compiles the wavefront-FFT channelizer's source itself -- csrc/channelizer_fft.hip, through tests/hostshim_wave64/hip/hip_runtime.h -- and runs its
// kernels with their wavefront semantics on the CPU (lanes as fibers; shuffles, barriers and the LDS exchanges of a wavefront as rendezvous points),
// launched by the file's own launch_channelizer_fft().  The test compares the bins it leaves in the stage-1 rings (and the last hop's spectrum) with a
// float64 FFT of the same converted, windowed samples.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../rtlsdr-airband_amd/csrc/channelizer_fft.hip"
#include "../rtlsdr-airband_amd/csrc/params.h"

using namespace airband;

extern "C" {

// iq: [n_dev] spans of iq_stride bytes.  out_mag [total_ch][n_hops] (AM channels; others 0), out_iq [total_ch][n_hops][2] (channels that need raw I/Q),
// spectrum [n_dev][2 * fft_size] (last hop) or null, window_out [fft_size].  bins_override [total_ch] or null: the bin every channel is tuned to.
int hostfft_run(const airband_hip_config* cfg, const uint8_t* iq, long iq_stride, int n_hops, int spectrum_only, const int* bins_override, float* out_mag, float* out_iq,
                float* spectrum, float* window_out) {
    Plan p;
    const int rc = build_plan(cfg, p);
    if (rc != 0) return rc;
    const int n_slots = (p.total_ch + 63) / 64 * 64;
    std::vector<ChanConst> cc(n_slots);
    std::vector<ChanState> cs(n_slots);
    std::memset(cc.data(), 0, sizeof(ChanConst) * n_slots);
    std::memset(cs.data(), 0, sizeof(ChanState) * n_slots);
    std::vector<int> ext_to_slot(p.total_ch);
    for (int e = 0; e < p.total_ch; e++) {
        cc[e] = p.cc[e];
        cs[e] = p.cs0[e];
        if (bins_override) cs[e].bin = bins_override[e];
        ext_to_slot[e] = e;
    }
    std::vector<DevConst> dev = p.dev;
    for (int d = 0; d < p.n_dev; d++) dev[d].chan_base = p.chan_base[d];
    for (auto& d : dev) d.any_afc = 1; /* every dongle's last-hop spectrum is wanted here (the library asks for it only where a channel has AFC) */
    const int R = (n_hops + 15) / 16 * 16;
    std::vector<float> mag((size_t)R * n_slots, 0.0f);
    std::vector<float2> bins((size_t)R * n_slots, make_float2(0.0f, 0.0f));
    ChannelizerArgs a;
    std::memset(&a, 0, sizeof(a));
    a.iq = iq;
    a.iq_stride = iq_stride;
    a.dev = dev.data();
    a.cs = cs.data();
    a.cc = cc.data();
    a.ext_to_slot = ext_to_slot.data();
    a.window = p.window.data();
    std::vector<float> wdec; /* as airband_hip_prepare() builds it */
    if (p.fft_size >= 1024) {
        const int M = p.fft_size / 512;
        wdec.resize(p.fft_size);
        for (int n = 0; n < p.fft_size; n++) wdec[(size_t)(n % M) * 512 + n / M] = p.window[n];
    }
    a.window_dec = wdec.empty() ? nullptr : wdec.data();
    a.twiddle = reinterpret_cast<const float2*>(p.twiddle.data());
    a.mag = mag.data();
    a.iq_bins = bins.data();
    a.last_spectrum = spectrum;
    a.n_dev = p.n_dev;
    a.fft_log = p.fft_log;
    a.hop_samples = p.dev[0].hop_samples;
    a.bytes_per_sample = p.dev[0].bytes_per_sample;
    a.sfmt = p.dev[0].sfmt;
    a.scale = p.dev[0].scale;
    a.row0 = 0;
    a.ring_rows = R;
    a.first_row = 0;
    a.n_hops = n_hops;
    a.max_ch = p.max_ch;
    a.spectrum_only = spectrum_only;
    if (fft_lds_bytes(a.fft_log, a.hop_samples, a.bytes_per_sample) > 160 * 1024) return -200;
    launch_channelizer_fft(a, nullptr);
    for (int e = 0; e < p.total_ch; e++)
        for (int t = 0; t < n_hops; t++) {
            const long off = ab_tile_base(e, R / AB_TILE_ROWS) + ab_tile_off(t);
            out_mag[(long)e * n_hops + t] = mag[off];
            out_iq[((long)e * n_hops + t) * 2] = bins[off].x;
            out_iq[((long)e * n_hops + t) * 2 + 1] = bins[off].y;
        }
    for (int i = 0; i < p.fft_size; i++) window_out[i] = p.window[i];
    return 0;
}

int hostfft_hop_samples(const airband_hip_config* cfg) {
    Plan p;
    return build_plan(cfg, p) != 0 ? -1 : p.dev[0].hop_samples;
}
}
