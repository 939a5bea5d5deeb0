/* csrc/params.cpp -- turns the user-level configuration into kernel constants.
 *
 * The reference does all of this at config-parse time, in double/float host arithmetic; the results are
 * plain numbers the hot loop then only reads.  Each block cites the reference lines whose *behaviour*
 * (formula, operand types, evaluation order) it reproduces, because parity of squelch decisions needs the
 * very same float constants.
 */
#include "params.h"
#include "exact_math.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <map>

namespace airband {

namespace {

const float kStandardTones[51] = {67.0f,  69.3f,  71.9f,  74.4f,  77.0f,  79.7f,  82.5f,  85.4f,  88.5f,  91.5f,  94.8f,  97.4f,  100.0f,
                                  103.5f, 107.2f, 110.9f, 114.8f, 118.8f, 123.0f, 127.3f, 131.8f, 136.5f, 141.3f, 146.2f, 150.0f, 151.4f,
                                  156.7f, 159.8f, 162.2f, 165.5f, 167.9f, 171.3f, 173.8f, 177.3f, 179.9f, 183.5f, 186.2f, 189.9f, 192.8f,
                                  196.6f, 199.5f, 203.5f, 206.5f, 210.7f, 218.1f, 225.7f, 229.1f, 233.6f, 241.8f, 250.3f, 254.1f};

/* Goertzel coefficient: nearest DFT bin of the window to the tone (reference: src/ctcss.cpp:31-42). */
float tone_coeff(float tone, float rate, int window) {
    const float bin_f = (float)window * tone / rate;
    const int k = (int)(0.5 + (double)bin_f);
    const float omega = (float)((2.0 * M_PI * k) / window);
    return (float)(2.0 * std::cos((double)omega));
}

/* Bank = target tone first, then the standard tones at least 5 Hz away, without duplicate coefficients
 * (reference: src/ctcss.cpp:61-73,105-122). */
void build_bank(float target, float rate, int window, float* coeff, int& n) {
    n = 0;
    auto add = [&](float f) {
        const float c = tone_coeff(f, rate, window);
        for (int i = 0; i < n; i++)
            if (coeff[i] == c) return;
        coeff[n++] = c;
    };
    add(target);
    for (float t : kStandardTones) {
        if (std::fabs(target - t) < 5) continue;
        add(t);
    }
}

float tau_alpha(int wave_rate, int tau_us) { /* src/config.cpp:648,775; src/rtl_airband.cpp:825-826 */
    return tau_us == 0 ? 0.0f : (float)std::exp(-1.0f / (wave_rate * 1e-6 * tau_us));
}

float dbfs_to_level(float dbfs, int fft_size) { /* src/util.cpp:169-176 */
    const float offset = 7.54f + 10.0f * log10f((float)(fft_size / 2)) - 2.38f;
    return (float)(std::pow(10.0, (dbfs - offset) / 20.0f) * fft_size);
}

/* 2nd-order Bessel lowpass via bilinear transform (reference: src/filters.cpp:70-144) */
void design_lowpass(float cutoff, float rate, float& gain, float& yc0, float& yc1) {
    typedef std::complex<double> cd;
    const double raw_alpha = (double)cutoff / rate;
    const double warped = std::tan(M_PI * raw_alpha) / M_PI;
    const cd bessel(-1.10160133059e+00, 6.36009824757e-01);
    auto blt = [](cd p) { return (2.0 + p) / (2.0 - p); };
    const cd roots_top[2] = {cd(-1.0), cd(-1.0)};
    const cd roots_bot[2] = {blt(M_PI * 2 * warped * bessel), blt(M_PI * 2 * warped * std::conj(bessel))};
    auto expand = [](const cd r[2], cd c[3]) {
        c[0] = 1.0;
        c[1] = 0.0;
        c[2] = 0.0;
        for (int i = 0; i < 2; i++) {
            const cd nw = -r[i];
            for (int k = 2; k >= 1; k--) c[k] = (nw * c[k]) + c[k - 1];
            c[0] = nw * c[0];
        }
    };
    auto eval = [](const cd c[3], cd z) {
        cd sum(0.0);
        for (int i = 2; i >= 0; i--) sum = (sum * z) + c[i];
        return sum;
    };
    cd top[3], bot[3];
    expand(roots_top, top);
    expand(roots_bot, bot);
    const cd g = eval(top, cd(1.0)) / eval(bot, cd(1.0));
    gain = (float)hypot(g.imag(), g.real());
    yc0 = (float)(-(bot[0].real() / bot[2].real()));
    yc1 = (float)(-(bot[1].real() / bot[2].real()));
}

}  // namespace

/* exact_math.h, ab_div_const_core(): the reciprocal to hand the kernels for divisor g, or 0 if the three-instruction division is not the IEEE one
 * for every dividend.  Every significand of one binade of x is tried (the operations scale exactly with x's exponent inside the range the
 * kernels use them in); sign symmetry holds operation by operation. */
float div_const_reciprocal(float g) {
    if (!(g >= 0x1p-40f && g <= 0x1p40f)) return 0.0f; /* (also NaN) keeps q, e and their products far from the ends of the exponent range */
    const float r = (float)(1.0 / (double)g);          /* double division, one more rounding: RN(1 / g) (1 / g is never within 2^-53 of a float midpoint) */
    for (uint32_t m = 0; m < (1u << 23); m++) {
        const float x = ab_float(0x3f800000u | m);
        if (ab_bits(ab_div_const_core(x, g, r)) != ab_bits(x / g)) return 0.0f;
    }
    return r;
}

namespace {

int fail(Plan& p, int code, const std::string& msg) {
    p.error = msg;
    return code;
}

}  // namespace

int build_plan(const airband_hip_config* cfg, Plan& p) {
    if (!cfg) return fail(p, AIRBAND_HIP_EINVAL, "config is NULL");
    if (cfg->abi_version != AIRBAND_HIP_ABI_VERSION) return fail(p, AIRBAND_HIP_EINVAL, "abi_version mismatch");
    if (cfg->fft_size_log < 8 || cfg->fft_size_log > 13) /* MIN/MAX_FFT_SIZE_LOG, src/rtl_airband.h:80-82 */
        return fail(p, AIRBAND_HIP_EBADSIZE, "fft_size_log must be between 8 and 13");
    if (cfg->wave_rate != 8000 && cfg->wave_rate != 16000) return fail(p, AIRBAND_HIP_EBADSIZE, "wave_rate must be 8000 (AM build) or 16000 (NFM build)");
    if (cfg->device_count < 1 || !cfg->devices) return fail(p, AIRBAND_HIP_EBADSIZE, "device_count must be >= 1");
    if (cfg->fm_demod != AIRBAND_FM_FAST_ATAN2 && cfg->fm_demod != AIRBAND_FM_QUADRI_DEMOD) return fail(p, AIRBAND_HIP_EINVAL, "unknown fm_demod");

    std::map<uint32_t, float> rgain_of; /* lowpass gain (bits) -> div_const_reciprocal(): a plan has a handful of distinct gains, the check takes ~70 ms each */
    p.fft_log = cfg->fft_size_log;
    p.fft_size = 1 << cfg->fft_size_log;
    p.wave_rate = cfg->wave_rate;
    p.wave_batch = cfg->wave_rate / 8;
    p.fm_demod = cfg->fm_demod;
    p.n_dev = cfg->device_count;

    /* sample -> float tables (src/rtl_airband.cpp:316-324); s8 entry 128 is never written by the reference */
    for (int i = 0; i < 256; i++) p.lev_u8[i] = (i - 127.5f) / 127.5f;
    /* the reference's loop leaves the entry of the byte -128 uninitialised (src/rtl_airband.cpp:322-324); the library continues the table's rule */
    for (int i = -128; i < 128; i++) p.lev_s8[(uint8_t)i] = i / 128.0f;

    /* window (src/rtl_airband.cpp:335-351): float literals widened to double, evaluated in double */
    {
        const double a0 = 0.27105140069342f, a1 = 0.43329793923448f, a2 = 0.21812299954311f, a3 = 0.06592544638803f, a4 = 0.01081174209837f,
                     a5 = 0.00077658482522f, a6 = 0.00001388721735f;
        p.window.resize(p.fft_size);
        const double den = (double)(p.fft_size - 1);
        for (int i = 0; i < p.fft_size; i++) {
            const double x = a0 - (a1 * cos((2.0 * M_PI * i) / den)) + (a2 * cos((4.0 * M_PI * i) / den)) - (a3 * cos((6.0 * M_PI * i) / den)) +
                             (a4 * cos((8.0 * M_PI * i) / den)) - (a5 * cos((10.0 * M_PI * i) / den)) + (a6 * cos((12.0 * M_PI * i) / den));
            p.window[i] = (float)x;
        }
    }
    /* derotation LUT (src/util.cpp:105-110) */
    p.sin_lut.resize(257);
    p.cos_lut.resize(257);
    for (uint32_t i = 0; i < 256; i++) sincosf((float)(2.0F * M_PI * (float)i / 256.0f), &p.sin_lut[i], &p.cos_lut[i]);
    p.sin_lut[256] = p.sin_lut[0];
    p.cos_lut[256] = p.cos_lut[0];
    /* twiddles of the wavefront-FFT channelizer: W_N^k = exp(-2 pi i k / N), evaluated in double */
    p.twiddle.resize((size_t)2 * p.fft_size);
    for (int k = 0; k < p.fft_size; k++) {
        p.twiddle[2 * k] = (float)std::cos(-2.0 * M_PI * (double)k / (double)p.fft_size);
        p.twiddle[2 * k + 1] = (float)std::sin(-2.0 * M_PI * (double)k / (double)p.fft_size);
    }

    const float global_alpha = (float)std::exp(-1.0f / (p.wave_rate * 2e-4)); /* src/rtl_airband.cpp:87 */
    const float rate = (float)p.wave_rate;

    p.dev.resize(p.n_dev);
    p.chan_base.resize(p.n_dev);
    p.total_ch = 0;
    p.max_ch = 0;
    for (int d = 0; d < p.n_dev; d++) {
        const airband_hip_device_cfg& dc = cfg->devices[d];
        DevConst& dev = p.dev[d];
        std::memset(&dev, 0, sizeof(dev));
        if (dc.channel_count < 1 || dc.channel_count > AB_MAX_CH_PER_DEV || !dc.channels)
            return fail(p, AIRBAND_HIP_EBADSIZE, "channel_count must be 1.." + std::to_string(AB_MAX_CH_PER_DEV));
        if (dc.sample_rate <= p.wave_rate) return fail(p, AIRBAND_HIP_EBADSIZE, "sample_rate must exceed WAVE_RATE"); /* src/config.cpp:790 */
        float fullscale;
        switch (dc.sfmt) { /* src/input-file.cpp:171-173, src/input-soapysdr.cpp:45-64 */
            case AIRBAND_SFMT_U8:
            case AIRBAND_SFMT_S8: dev.bytes_per_sample = 1; fullscale = (float)SCHAR_MAX - 0.5f; break;
            case AIRBAND_SFMT_S16: dev.bytes_per_sample = 2; fullscale = (float)SHRT_MAX - 0.5f; break;
            case AIRBAND_SFMT_F32: dev.bytes_per_sample = 4; fullscale = 1.0f; break;
            default: return fail(p, AIRBAND_HIP_EINVAL, "unknown sample format");
        }
        if (dc.fullscale > 0) fullscale = dc.fullscale;
        dev.sfmt = dc.sfmt;
        dev.scale = 1.0f / fullscale;
        dev.hop_samples = (int)std::round((double)dc.sample_rate / (double)p.wave_rate); /* src/rtl_airband.cpp:394 */
        dev.n_ch = dc.channel_count;
        dev.chan_base = p.total_ch;
        p.chan_base[d] = p.total_ch;
        const int64_t hop_bytes = 2LL * dev.bytes_per_sample * dev.hop_samples;
        if (d > 0 && (dev.sfmt != p.dev[0].sfmt || dev.hop_samples != p.dev[0].hop_samples)) p.uniform_hop = false;
        if (hop_bytes > p.hop_bytes_max) p.hop_bytes_max = hop_bytes;
        const float dev_alpha = dc.tau_us >= 0 ? tau_alpha(p.wave_rate, dc.tau_us) : global_alpha; /* src/config.cpp:774-778 */

        for (int j = 0; j < dc.channel_count; j++) {
            const airband_hip_channel_cfg& ch = dc.channels[j];
            ChanConst c;
            ChanState s;
            std::memset(&c, 0, sizeof(c));
            std::memset(&s, 0, sizeof(s));
            c.flags = AB_F_VALID;
            c.dev = d;
            c.chan = j;
            c.ext_index = p.total_ch;
            c.afc = ch.afc & 0xff;
            if (c.afc) dev.any_afc = 1;
            c.ct_slot = -1;
            if (ch.modulation == AIRBAND_MOD_NFM) {
                if (p.wave_rate != 16000) return fail(p, AIRBAND_HIP_EINVAL, "NFM channels need wave_rate 16000 (the reference's NFM build)");
                c.flags |= AB_F_NFM | AB_F_RAW_IQ; /* src/config.cpp:670-677 */
                if (p.fm_demod == AIRBAND_FM_QUADRI_DEMOD) c.flags |= AB_F_QUADRI;
            } else if (ch.modulation != AIRBAND_MOD_AM) {
                return fail(p, AIRBAND_HIP_EINVAL, "unknown modulation");
            }
            if (ch.ampfactor < 0) return fail(p, AIRBAND_HIP_EINVAL, "ampfactor must not be negative"); /* src/config.cpp:624-627 */
            c.ampfactor = ch.ampfactor;
            c.alpha = ch.tau_us >= 0 ? tau_alpha(p.wave_rate, ch.tau_us) : dev_alpha;

            /* Squelch defaults (src/squelch.cpp:36-77) then the two threshold knobs in config order (src/config.cpp:452-515) */
            s.noise_floor = 5.0f;
            c.sq_normal_ratio = (float)std::pow(10.0, 9.54f / 20.0);
            c.sq_manual_level = -1.0f;
            bool manual = false;
            if (ch.squelch_threshold_dbfs > 0) return fail(p, AIRBAND_HIP_EINVAL, "squelch_threshold must be <= 0"); /* src/config.cpp:457 */
            if (ch.squelch_threshold_dbfs < 0) {
                const float lvl = dbfs_to_level((float)ch.squelch_threshold_dbfs, p.fft_size);
                if (lvl > 0) { /* Squelch::set_squelch_level_threshold, src/squelch.cpp:79-91 */
                    manual = true;
                    c.sq_manual_level = lvl;
                }
            }
            if (ch.squelch_snr_threshold_db >= 0.0f) { /* Squelch::set_squelch_snr_threshold, src/squelch.cpp:93-103 */
                manual = false;
                c.sq_normal_ratio = (float)std::pow(10.0, ch.squelch_snr_threshold_db / 20.0);
            } else if (ch.squelch_snr_threshold_db != -1.0f) {
                return fail(p, AIRBAND_HIP_EINVAL, "squelch_snr_threshold must be >= 0 (or -1 for the default)"); /* src/config.cpp:494-497 */
            }
            c.sq_flappy_ratio = c.sq_normal_ratio * 0.9f;
            if (manual) c.flags |= AB_F_MANUAL;
            s.cap = manual ? 1.5f * c.sq_manual_level : 1.5f * c.sq_normal_ratio * s.noise_floor; /* src/squelch.cpp:492-499 */
            s.pre_full = s.pre_capped = s.post_full = s.post_capped = 0.001f;
            s.sh_nf = s.noise_floor; s.sh_cap = s.cap; s.sh_capped = s.pre_capped; /* the same machine, 101 samples behind */
            s.row_zero = 2; /* result rows start as zeros behind a carry prefilled with 0.5 (airband_hip.cpp, src/config.cpp:313-316) */
            s.level_cache = 0.0f;
            s.next = s.cur = AB_ST_CLOSED;
            s.sample_count = 0xffffffffu;
            s.head = 0;
            s.tail = 1;

            if (ch.notch_freq > 0) { /* NotchFilter ctor, src/filters.cpp:30-48; default q src/config.cpp:517 */
                const float q = ch.notch_q > 0 ? ch.notch_q : 10.0f;
                const float wo = (float)(2 * M_PI * (double)(ch.notch_freq / rate));
                const float e = 1 / (1 + std::tan(wo / (q * 2)));
                const float pc = std::cos(wo);
                c.notch_d0 = e;
                c.notch_d1 = 2 * e * pc;
                c.notch_d2 = (2 * e - 1);
                c.flags |= AB_F_NOTCH;
            }
            if (ch.ctcss_freq > 0) { /* Squelch::set_ctcss_freq, src/squelch.cpp:105-116 */
                ToneTable t;
                std::memset(&t, 0, sizeof(t));
                t.window[0] = (int)(rate * 0.05);
                t.window[1] = (int)(rate * 0.4);
                for (int k = 0; k < 2; k++) build_bank(ch.ctcss_freq, rate, t.window[k], t.coeff[k], t.n[k]);
                c.ct_slot = (int)p.tones.size();
                for (int k = 0; k < 2; k++) {
                    c.ct_ntones[k] = t.n[k];
                    c.ct_window[k] = t.window[k];
                }
                p.tones.push_back(t);
                c.flags |= AB_F_CTCSS;
            }
            if (ch.bandwidth_hz != 0) { /* src/config.cpp:592-619 */
                c.flags |= AB_F_RAW_IQ;
                if (ch.bandwidth_hz > 0) {
                    design_lowpass((float)ch.bandwidth_hz / 2, rate, c.lp_gain, c.lp_yc0, c.lp_yc1);
                    c.flags |= AB_F_LOWPASS;
                    auto known = rgain_of.find(ab_bits(c.lp_gain));
                    if (known == rgain_of.end()) known = rgain_of.emplace(ab_bits(c.lp_gain), div_const_reciprocal(c.lp_gain)).first;
                    c.lp_rgain = known->second;
                }
            }
            if (ch.has_iq_outputs) c.flags |= AB_F_IQ_OUT | AB_F_RAW_IQ;

            /* bin index (src/config.cpp:666-667): the divisor is the INTEGER quotient sample_rate / fft_size */
            const double pos = (ch.frequency + dc.sample_rate - dc.centerfreq) / (double)(dc.sample_rate / p.fft_size) - 1.0;
            c.base_bin = (int)((size_t)std::ceil(pos) % (size_t)p.fft_size);
            s.bin = c.base_bin;

            if (c.flags & AB_F_RAW_IQ) { /* derotation step (src/config.cpp:679-712) */
                double f = (double)(ch.frequency - dc.centerfreq);
                const double dec = (double)dc.sample_rate / (double)p.wave_rate;
                double corr = (double)p.wave_rate / 2.0;
                corr *= (dec - std::round(dec));
                corr *= (double)(ch.frequency - dc.centerfreq) / ((double)dc.sample_rate / 2.0);
                f -= corr;
                f /= (double)p.wave_rate;
                f -= std::trunc(f);
                f *= 256.0 * 65536.0;
                c.dm_dphi = (uint32_t)((int)f);
                dev.any_raw_iq = 1;
            }
            /* freq_t / channel_t initial values (src/config.cpp:274,313-331) */
            s.agcavgfast = 0.5f;
            s.pr = s.pj = 0.0f;
            s.prev_waveout = 0.5f;
            s.axc = ' ';
            p.cc.push_back(c);
            p.cs0.push_back(s);
            p.total_ch++;
        }
        if (dc.channel_count > p.max_ch) p.max_ch = dc.channel_count;
    }
    return AIRBAND_HIP_OK;
}

/* Coefficient table of the pruned DFT: for window byte k = 2n + {0: I, 1: Q} and column 2c + {0: re, 1: im}
 *   (I + jQ) * w[n] * exp(-j theta) = (I w cos + Q w sin) + j (Q w cos - I w sin),  theta = 2 pi bin_c n / N
 * (same transform as src/rtl_airband.cpp:451-460 + :483-489 evaluated for one bin).  Values are scaled to 24-bit
 * integers and split into balanced base-256 digits d0 + 256 d1 + 65536 d2, each in [-128, 127]. */
void build_f32_tables(Plan& p) {
    /* fft_size 4096 / 8192: the window in segments of 2 048 samples, each laid out like an fft 2048 table (kernels.h f32_seg_size / f32_n_seg) */
    const int N = p.fft_size, SEG = N < 2048 ? N : 2048, NSEG = N / SEG;
    const int NW = N <= 512 ? 4 : 8, VPP = 2 * SEG / NW, KW = VPP / 4; /* values and MFMAs (K = 4) per piece; NW = kernels.h f32_nw() */
    p.ftab.assign((size_t)p.n_shared_bsets * NSEG * NW * KW * 64, 0.0f);
    for (int b = 0; b < p.n_shared_bsets; b++)
      for (int seg = 0; seg < NSEG; seg++)
        for (int piece = 0; piece < NW; piece++)
            for (int s = 0; s < KW; s++)
                for (int lane = 0; lane < 64; lane++) {
                    const int col = lane & 15, c = col >> 1, g = lane >> 4;
                    const int bin = p.bset_bins[(size_t)b * 8 + c];
                    if (bin < 0) continue; /* a group with fewer than 8 channels: unused columns stay zero */
                    const int k = seg * 2 * SEG + piece * VPP + 16 * (s / 4) + 4 * g + (s % 4);
                    const int n = k >> 1;
                    const double th = 2.0 * M_PI * (double)(((long long)bin * n) % N) / (double)N; /* the phase is reduced exactly in integers first */
                    const double wc = (double)p.window[n] * std::cos(th), ws = (double)p.window[n] * std::sin(th);
                    /* (I + jQ) w e^{-j th} = (I w cos + Q w sin) + j (Q w cos - I w sin): value k = 2n is I, 2n + 1 is Q; column 2c is re, 2c + 1 im */
                    const double v = (col & 1) ? ((k & 1) ? wc : -ws) : ((k & 1) ? ws : wc);
                    p.ftab[((((size_t)b * NSEG + seg) * NW + piece) * KW + s) * 64 + lane] = (float)v;
                }
}

/* shared coefficient tables the host builds; the rest are built on the device at prepare() time.  AIRBAND_HIP_HOST_TABLES_MAX overrides (tests: small fleets through the
 * device builder) */
static int host_tables_max() {
    static const int n = [] {
        const char* e = getenv("AIRBAND_HIP_HOST_TABLES_MAX");
        const long v = e ? strtol(e, nullptr, 10) : 4096;
        return (int)(v < 0 ? 0 : v > (1 << 20) ? (1 << 20) : v);
    }();
    return n;
}

void build_dft_tables(Plan& p, bool host_private) {
    /* fft_size > 512: one table per window piece of 512 samples (NP pieces); a table then covers K = 1024 bytes of the window */
    const int N = p.fft_size, NP = N > 512 ? N / 512 : 1, NS = N / NP, K = 2 * NS, KS = K / 64;
    const double S = AB_DFT_COEF_SCALE;
    p.b_unscale = 1.0 / (S * 127.5);
    p.item_dev.clear();
    p.item_group.clear();
    p.item_bset.clear();
    p.item_home.clear();
    p.bfrag.clear();
    p.bcorr.clear();
    /* Tables are SHARED between work items with the same bins (a fleet of identical dongles has one).  A group with an AFC channel ALSO owns
     * a private table: AFC moves the channel's bin at run time (src/rtl_airband.cpp:224-250), the re-tune kernel then rewrites that channel's
     * two columns there -- and points the work item at the private table only while some channel of the group is away from its base bin
     * (item_home: the shared table of the base bins, which it reads otherwise; 65 536 private tables are 3.2 GB of coefficient reads per
     * batch).  Shared tables take the indices [0, n_shared), private ones follow; private tables are built by the re-tune kernel itself on
     * the device (host_private: and here, for the host-only self test). */
    std::vector<std::vector<int>> shared_keys, private_keys;
    auto build = [&](const std::vector<int>& key) {
        for (int piece = 0; piece < NP; piece++) {
            std::vector<int> q((size_t)K * 16, 0);
            for (int c = 0; c < (int)key.size(); c++) {
                for (int i = 0; i < NS; i++) {
                    const int n = piece * NS + i;
                    /* the phase is reduced exactly in integers before it meets a double */
                    const double th = 2.0 * M_PI * (double)(((long long)key[c] * n) % N) / (double)N;
                    const double wc = (double)p.window[n] * std::cos(th), ws = (double)p.window[n] * std::sin(th);
                    q[(size_t)(2 * i) * 16 + 2 * c] = (int)std::llround(wc * S);       /* I -> re */
                    q[(size_t)(2 * i + 1) * 16 + 2 * c] = (int)std::llround(ws * S);   /* Q -> re */
                    q[(size_t)(2 * i) * 16 + 2 * c + 1] = (int)std::llround(-ws * S);  /* I -> im */
                    q[(size_t)(2 * i + 1) * 16 + 2 * c + 1] = (int)std::llround(wc * S); /* Q -> im */
                }
            }
            const size_t base = p.bfrag.size();
            p.bfrag.resize(base + (size_t)3 * KS * 64 * 16, 0);
            for (int col = 0; col < 16; col++) {
                double sum = 0.0;
                for (int k = 0; k < K; k++) {
                    const int v = q[(size_t)k * 16 + col];
                    sum += v;
                    int dgt[3];
                    int rest = v;
                    for (int t = 0; t < 3; t++) {
                        int lo = ((rest + 128) & 255) - 128; /* balanced digit */
                        dgt[t] = lo;
                        rest = (rest - lo) / 256;
                    }
                    const int s = k / 64, gg = (k % 64) / 16, jj = k % 16;
                    const int lane = gg * 16 + col;
                    for (int t = 0; t < 3; t++) p.bfrag[base + (((size_t)t * KS + s) * 64 + lane) * 16 + jj] = (int8_t)dgt[t];
                }
                p.bcorr.push_back(0.5 * sum); /* (b - 127.5) = (b - 128) + 0.5 */
            }
        }
    };
    int last_shared = -1;
    /* every device_t derives its own bins (src/config.cpp:666-667): a fleet may hold as many distinct groups of eight bins as it has dongles.  The lookup is a
     * map (it was a linear search: 65 536 distinct plans = 2e9 vector compares), the host builds the first host_tables_max() = 4 096 tables and leaves the rest to the
     * device (misc_kernels.hip, build_tables_kernel, at prepare() time: same arithmetic, the device's sincospi) -- a 3 GB host image of 65 536 tables is not needed
     * to run them.  host_private (the host-only self test) builds everything here. */
    std::map<std::vector<int>, int> shared_index;
    for (int d = 0; d < p.n_dev; d++) {
        for (int g = 0; g * 8 < p.dev[d].n_ch; g++) { /* one work item per group of 8 channels */
            std::vector<int> key;
            bool private_table = false;
            for (int j = g * 8; j < p.dev[d].n_ch && j < g * 8 + 8; j++) {
                key.push_back(p.cc[p.chan_base[d] + j].base_bin);
                private_table |= p.cc[p.chan_base[d] + j].afc != 0;
            }
            int found = -1, home = -1;
            if (last_shared >= 0 && shared_keys[last_shared] == key) home = last_shared; /* fleets of identical dongles: the common case */
            if (home < 0) {
                auto it = shared_index.find(key);
                if (it != shared_index.end()) home = it->second;
            }
            if (home < 0) {
                home = (int)shared_keys.size();
                shared_keys.push_back(key);
                shared_index.emplace(key, home);
                if (host_private || shared_keys.size() <= (size_t)host_tables_max()) build(key); /* the rest: on the device */
            }
            last_shared = home;
            if (private_table) {
                found = -(int)private_keys.size() - 1; /* its final index is known once the shared tables are counted */
                private_keys.push_back(key);
            } else {
                found = home;
            }
            p.item_home.push_back(home);
            p.item_dev.push_back(d);
            p.item_group.push_back(g);
            p.item_bset.push_back(found);
        }
    }
    p.n_shared_bsets = (int)shared_keys.size();
    p.n_host_bsets = host_private ? p.n_shared_bsets : std::min(p.n_shared_bsets, host_tables_max());
    p.n_bsets = p.n_shared_bsets + (int)private_keys.size();
    for (int& b : p.item_bset)
        if (b < 0) b = p.n_shared_bsets + (-b - 1);
    p.bset_bins.assign((size_t)p.n_bsets * 8, -1);
    for (size_t b = 0; b < shared_keys.size(); b++)
        for (size_t c = 0; c < shared_keys[b].size(); c++) p.bset_bins[b * 8 + c] = shared_keys[b][c];
    if (host_private)
        for (size_t b = 0; b < private_keys.size(); b++) {
            build(private_keys[b]);
            for (size_t c = 0; c < private_keys[b].size(); c++) p.bset_bins[(p.n_shared_bsets + b) * 8 + c] = private_keys[b][c];
        }
    /* the window is below 2^-8 of full scale in the outer k-steps, so the top digit vanishes there: verify, don't assume */
    p.b_edge_hi_zero = NP == 1; /* window pieces have their small coefficients at one end only */
    const int edge = KS / 8;
    /* tables that AFC may re-tune to any bin: the top digit of an edge coefficient is zero whatever the bin iff the window itself is that small there */
    /* (the same for shared tables the device builds: they are not here to be looked at) */
    for (int i = 0; (!private_keys.empty() || p.n_host_bsets < p.n_shared_bsets) && p.b_edge_hi_zero && i < NS; i++) {
        const int s_ = (2 * i) / 64;
        if (s_ >= edge && s_ < KS - edge) continue;
        if ((double)p.window[i] * S > 32639.0) p.b_edge_hi_zero = false;
    }
    const int n_built = (int)(p.bfrag.size() / ((size_t)3 * KS * 64 * 16 * NP));
    for (int b = 0; b < n_built && p.b_edge_hi_zero; b++)
        for (int s = 0; s < KS; s++) {
            if (s >= edge && s < KS - edge) continue;
            for (int i = 0; i < 64 * 16; i++)
                if (p.bfrag[(size_t)b * 3 * KS * 64 * 16 + ((size_t)2 * KS + s) * 64 * 16 + i] != 0) p.b_edge_hi_zero = false;
        }
}

/* Host-only check of the coefficient tables: for pseudo-random raw windows, the value the matrix-core channelizer would produce -- the
 * integer sums of (byte - 128) x balanced base-256 digits over every window piece, recombined, offset-corrected and scaled exactly as
 * the kernel does -- against the definition  X[bin] = sum_n lev[b_n] w[n] exp(-2 pi i bin n / N)  (src/rtl_airband.cpp:316-351,402-489)
 * evaluated directly in double.  Returns the largest error relative to the RMS of the exact values over all work items. */
/* host-only: the float tables of channelizer_f32.hip contracted the way the kernel contracts them -- per wave (piece) 64 (32) instructions of K = 4, lane
 * l supplying stream value 16 (s / 4) + 4 (l >> 4) + s % 4 of its piece against table entry [piece][s][l], partial sums of a piece in float, the four pieces
 * added in float -- against the double-precision sum, on pseudo-random windows; largest error relative to the RMS of the exact values */
double f32_table_selftest(const Plan& p, int windows) {
    const int N = p.fft_size, SEG = N < 2048 ? N : 2048, NSEG = N / SEG, NW = N <= 512 ? 4 : 8, VPP = 2 * SEG / NW, KW = VPP / 4;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
        return rng;
    };
    double worst = 0.0, sumsq = 0.0;
    long count = 0;
    for (size_t item = 0; item < p.item_dev.size(); item++) {
        const int d = p.item_dev[item], g = p.item_group[item], set = p.item_home[item];
        const int nch = std::min(8, p.dev[d].n_ch - 8 * g);
        for (int w = 0; w < windows; w++) {
            std::vector<float> raw(2 * N);
            for (int k = 0; k < 2 * N; k++) raw[k] = (float)((double)(int32_t)(next() >> 32) / 2147483648.0 * 0.3);
            for (int c = 0; c < nch; c++) {
                const int bin = p.cc[p.chan_base[d] + 8 * g + c].base_bin;
                for (int comp = 0; comp < 2; comp++) {
                    const int col = 2 * c + comp;
                    float total = 0.0f;
                    for (int seg = 0; seg < NSEG; seg++) { /* one launch per window segment; the sums of the earlier segments are added to the finishing wave's */
                        float seg_total = 0.0f;
                        for (int piece = 0; piece < NW; piece++) {
                            float acc2[2] = {0.0f, 0.0f}; /* the kernel alternates two accumulators */
                            for (int s_ = 0; s_ < KW; s_++) {
                                float part = acc2[s_ & 1]; /* one MFMA: four products added to the accumulator */
                                for (int gg = 0; gg < 4; gg++)
                                    part += raw[seg * 2 * SEG + piece * VPP + 16 * (s_ / 4) + 4 * gg + (s_ % 4)] * p.ftab[((((size_t)set * NSEG + seg) * NW + piece) * KW + s_) * 64 + gg * 16 + col];
                                acc2[s_ & 1] = part;
                            }
                            const float acc = acc2[0] + acc2[1];
                            seg_total = piece == 0 ? acc : seg_total + acc;
                        }
                        total = seg == 0 ? seg_total : seg_total + total;
                    }
                    const double got = (double)(total * p.dev[d].scale);
                    double want = 0.0;
                    for (int n = 0; n < N; n++) {
                        const double th = 2.0 * M_PI * (double)(((long long)bin * n) % N) / (double)N;
                        const double xi = (double)p.dev[d].scale * raw[2 * n], xq = (double)p.dev[d].scale * raw[2 * n + 1], wn = (double)p.window[n];
                        want += comp == 0 ? wn * (xi * std::cos(th) + xq * std::sin(th)) : wn * (xq * std::cos(th) - xi * std::sin(th));
                    }
                    worst = std::max(worst, std::fabs(got - want));
                    sumsq += want * want;
                    count++;
                }
            }
        }
    }
    return count ? worst / std::sqrt(sumsq / (double)count) : 0.0;
}

double dft_table_selftest(const Plan& p, int windows) {
    const int N = p.fft_size, NP = N > 512 ? N / 512 : 1, NS = N / NP, K = 2 * NS, KS = K / 64;
    const bool s16 = p.dev[0].sfmt == AIRBAND_SFMT_S16, s8 = p.dev[0].sfmt == AIRBAND_SFMT_S8;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
        return rng;
    };
    double worst = 0.0, sumsq = 0.0;
    long count = 0;
    std::vector<double> errs;
    for (size_t item = 0; item < p.item_dev.size(); item++) {
        const int d = p.item_dev[item], g = p.item_group[item], set = p.item_bset[item];
        const int nch = std::min(8, p.dev[d].n_ch - 8 * g);
        for (int w = 0; w < windows; w++) {
            std::vector<int> raw(2 * N); /* u8: 0..255; s8: -128..127; CS16: -32768..32767 */
            for (int k = 0; k < 2 * N; k++) raw[k] = s16 ? (int)(int16_t)(next() >> 17) : s8 ? (int)(int8_t)((next() >> 23) & 255) : (int)((next() >> 23) & 255);
            for (int c = 0; c < nch; c++) {
                const int bin = p.cc[p.chan_base[d] + 8 * g + c].base_bin;
                for (int comp = 0; comp < 2; comp++) {
                    const int col = 2 * c + comp;
                    double table_units = 0.0; /* what the int32 accumulators + f64 recombination hold, all pieces */
                    for (int piece = 0; piece < NP; piece++) {
                        const size_t base = ((size_t)set * NP + piece) * 3 * KS * 64 * 16;
                        long long acc = 0;
                        for (int k = 0; k < K; k++) {
                            const int s_ = k / 64, gg = (k % 64) / 16, jj = k % 16, lane = gg * 16 + col;
                            long long coef = 0;
                            for (int t = 2; t >= 0; t--) coef = coef * 256 + p.bfrag[base + (((size_t)t * KS + s_) * 64 + lane) * 16 + jj];
                            const int v = raw[piece * K + k];
                            /* u8: (b - 128) + the 0.5 of the correction term; s8: the byte as it is; CS16: (lo - 128) + 256 hi + 128 = the sample itself */
                            acc += (long long)((s16 || s8) ? v : v - 128) * coef;
                        }
                        table_units += (double)acc + ((s16 || s8) ? 0.0 : p.bcorr[((size_t)set * NP + piece) * 16 + col]);
                    }
                    const double got = table_units * (s16 ? p.b_unscale * 127.5 * (double)p.dev[d].scale : s8 ? p.b_unscale * 127.5 / 128.0 : p.b_unscale);
                    double want = 0.0;
                    for (int n = 0; n < N; n++) {
                        const double th = 2.0 * M_PI * (double)(((long long)bin * n) % N) / (double)N;
                        const double xi = s16 ? (double)p.dev[d].scale * raw[2 * n] : s8 ? (double)p.lev_s8[(uint8_t)raw[2 * n]] : (double)p.lev_u8[raw[2 * n]];
                        const double xq = s16 ? (double)p.dev[d].scale * raw[2 * n + 1] : s8 ? (double)p.lev_s8[(uint8_t)raw[2 * n + 1]] : (double)p.lev_u8[raw[2 * n + 1]];
                        const double wn = (double)p.window[n];
                        want += comp == 0 ? wn * (xi * std::cos(th) + xq * std::sin(th)) : wn * (xq * std::cos(th) - xi * std::sin(th));
                    }
                    errs.push_back(std::fabs(got - want));
                    sumsq += want * want;
                    count++;
                }
            }
        }
    }
    const double rms = count ? std::sqrt(sumsq / (double)count) : 1.0;
    for (double e : errs) worst = std::max(worst, e / (rms > 0 ? rms : 1.0));
    return worst;
}

void channel_constants(const Plan& p, int i, double* v) {
    const ChanConst& c = p.cc[i];
    v[0] = c.base_bin;
    v[1] = c.dm_dphi;
    v[2] = c.alpha;
    v[3] = c.notch_d0;
    v[4] = c.notch_d1;
    v[5] = c.notch_d2;
    v[6] = c.lp_gain;
    v[7] = c.lp_yc0;
    v[8] = c.lp_yc1;
    v[9] = c.sq_normal_ratio;
    v[10] = (c.flags & AB_F_MANUAL) ? c.sq_manual_level : -1.0;
    v[11] = c.ct_slot >= 0 ? c.ct_ntones[0] : 0;
    v[12] = c.ct_slot >= 0 ? c.ct_ntones[1] : 0;
    v[13] = (c.flags & AB_F_RAW_IQ) ? 1 : 0;
    v[14] = c.ct_slot >= 0 ? c.ct_window[0] : 0;
    v[15] = c.ct_slot >= 0 ? c.ct_window[1] : 0;
}

}  // namespace airband
