// demod_hip.cpp -- RTLSDR-Airband side of the MI355X backend: demodulate() for a shard of devices, done by
// libairband_hip.so (C ABI: airband_hip.h).  Added to the reference tree by integration/airband_hip.patch together with
// the few fields / one setter it uses; everything else -- main(), the config parser, the input drivers and their ring
// buffers, output / mixer threads, the stats file -- is untouched.
//
// Selected like the VideoCore FFT backend (WITH_BCM_VC, rtl_airband.cpp:293-310): a compile-time switch, here
// WITH_AIRBAND_HIP, which makes main() start demodulate_hip() instead of demodulate() (rtl_airband.cpp:1110-1112).
#include <pthread.h>
#include <string.h>
#include <unistd.h>

#include <libconfig.h++>
#include <vector>

#include "airband_hip.h"
#include "rtl_airband.h"

#ifdef NFM
// the -Q command-line switch lives in rtl_airband.cpp only (rtl_airband.cpp:88-89); same declaration here
enum fm_demod_algo { FM_FAST_ATAN2, FM_QUADRI_DEMOD };
extern enum fm_demod_algo fm_demod;
#endif

// The config-level numbers of a channel (freqlist[0]: multichannel mode), remembered next to the objects parse_channels()
// builds from them (config.cpp:436-650).  Lists are the per-frequency form of scan mode; their first element is ours.
static libconfig::Setting& first_of(libconfig::Setting& s) {
    return s.getType() == libconfig::Setting::TypeList ? s[0] : s;
}
static float num_of(libconfig::Setting& s) {
    libconfig::Setting& v = first_of(s);
    return v.getType() == libconfig::Setting::TypeFloat ? (float)v : (float)(int)v;
}
void demod_hip_keep_channel_cfg(libconfig::Setting& chan, channel_t* channel) {
    channel->cfg_squelch_threshold = chan.exists("squelch_threshold") ? (int)num_of(chan["squelch_threshold"]) : 0;
    channel->cfg_squelch_snr = chan.exists("squelch_snr_threshold") ? num_of(chan["squelch_snr_threshold"]) : -1.0f;
    channel->cfg_notch = chan.exists("notch") ? num_of(chan["notch"]) : 0.0f;
    channel->cfg_notch_q = chan.exists("notch_q") ? num_of(chan["notch_q"]) : 0.0f;
    channel->cfg_ctcss = chan.exists("ctcss") ? num_of(chan["ctcss"]) : 0.0f;
    channel->cfg_bandwidth = chan.exists("bandwidth") ? (int)num_of(chan["bandwidth"]) : 0;
    channel->cfg_tau = chan.exists("tau") ? (int)chan["tau"] : -1;
    if (channel->cfg_notch < 0) channel->cfg_notch = 0;  // "invalid, ignoring" (config.cpp:541,556)
    if (channel->cfg_ctcss < 0) channel->cfg_ctcss = 0;  // config.cpp:575,584
    if (channel->cfg_bandwidth < 0) channel->cfg_bandwidth = 0;  // config.cpp:601,609
}

static airband_hip_channel_cfg channel_cfg_of(const channel_t* ch) {
    const freq_t* f = ch->freqlist;  // multichannel mode: freq_count == 1
    airband_hip_channel_cfg c;
    memset(&c, 0, sizeof(c));
    c.frequency = f->frequency;
    c.modulation = f->modulation == MOD_AM ? AIRBAND_MOD_AM : AIRBAND_MOD_NFM;
    c.afc = ch->afc;
    c.squelch_threshold_dbfs = ch->cfg_squelch_threshold;
    c.squelch_snr_threshold_db = ch->cfg_squelch_snr;
    c.notch_freq = ch->cfg_notch;
    c.notch_q = ch->cfg_notch_q;
    c.ctcss_freq = ch->cfg_ctcss;
    c.bandwidth_hz = ch->cfg_bandwidth;
    c.ampfactor = f->ampfactor;
    c.tau_us = ch->cfg_tau;
    c.has_iq_outputs = ch->has_iq_outputs;
    return c;
}

// One library handle per CLASS of devices: libairband_hip batches the dongles of a handle through one launch, so they must share sample
// format and hop (round(sample_rate / WAVE_RATE), rtl_airband.cpp:394); the reference takes both per device (input-common.h:39-57).
struct hip_class {
    airband_hip_handle* h;
    std::vector<int> devs;  // indices into devices[]
    airband_hip_geometry g;
    std::vector<float> wave, iq;
    std::vector<char> axc;
    std::vector<airband_hip_channel_stats> st;
};

static bool same_class(const input_t* a, const input_t* b) {
    return a->sfmt == b->sfmt && a->bytes_per_sample == b->bytes_per_sample &&
           round((double)a->sample_rate / (double)WAVE_RATE) == round((double)b->sample_rate / (double)WAVE_RATE);
}

static void prepare_class(hip_class& k, int hip_device) {
    const int n = (int)k.devs.size();
    std::vector<std::vector<airband_hip_channel_cfg> > ch(n);
    std::vector<airband_hip_device_cfg> dv(n);
    for (int i = 0; i < n; i++) {
        device_t* dev = devices + k.devs[i];
        for (int j = 0; j < dev->channel_count; j++) ch[i].push_back(channel_cfg_of(dev->channels + j));
        memset(&dv[i], 0, sizeof(dv[i]));
        dv[i].sample_rate = dev->input->sample_rate;
        dv[i].centerfreq = dev->input->centerfreq;
        dv[i].sfmt = (int32_t)dev->input->sfmt;
        dv[i].fullscale = dev->input->fullscale;
        dv[i].tau_us = dev->cfg_tau;
        dv[i].channel_count = dev->channel_count;
        dv[i].channels = ch[i].data();
    }
    airband_hip_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = AIRBAND_HIP_ABI_VERSION;
    cfg.fft_size_log = (int32_t)fft_size_log;
    cfg.wave_rate = WAVE_RATE;
#ifdef NFM
    cfg.fm_demod = fm_demod == FM_QUADRI_DEMOD ? AIRBAND_FM_QUADRI_DEMOD : AIRBAND_FM_FAST_ATAN2;
#endif
    cfg.hip_device = hip_device;
    cfg.device_count = n;
    cfg.devices = dv.data();
    k.h = NULL;
    switch (airband_hip_prepare(&cfg, &k.h)) {  // the convention of gpu_fft_prepare() (rtl_airband.cpp:296-310)
        case AIRBAND_HIP_OK:
            break;
        case AIRBAND_HIP_ENODEV:
            log(LOG_CRIT, "No usable HIP device: %s\n", airband_hip_last_error(NULL));
            error();
            break;
        case AIRBAND_HIP_EBADSIZE:
            log(LOG_CRIT, "fft_size_log=%zu or sample rate not supported by airband_hip: %s\n", fft_size_log, airband_hip_last_error(NULL));
            error();
            break;
        case AIRBAND_HIP_ENOMEM:
            log(LOG_CRIT, "Out of GPU memory\n");
            error();
            break;
        default:
            log(LOG_CRIT, "airband_hip: %s\n", airband_hip_last_error(NULL));
            error();
    }
    airband_hip_get_geometry(k.h, &k.g);
    k.wave.resize((size_t)k.g.total_channels * k.g.wave_batch);
    k.iq.resize((size_t)k.g.total_channels * k.g.wave_batch * 2);
    k.axc.resize(k.g.total_channels);
    k.st.resize(k.g.total_channels);
}

void* demodulate_hip(void* params) {
    demod_params_t* dp = (demod_params_t*)params;  // device_start / device_end shard (rtl_airband.h demod_params_t)
    // one GPU per shard: with multiple_demod_threads a shard is one device (rtl_airband.cpp:1052-1086), spread round robin
    const int gpus = airband_hip_gpu_count();
    const int hip_device = gpus > 0 ? dp->device_start % gpus : 0;
    std::vector<hip_class> classes;
    for (int d = dp->device_start; d < dp->device_end; d++) {
        if (devices[d].mode != R_MULTICHANNEL) {  // scan mode retunes the dongle between batches (rtl_airband.cpp:556-565 of the controller thread)
            log(LOG_CRIT, "airband_hip: device %d is in scan mode; the GPU backend demodulates multichannel devices only\n", d);
            error();
        }
        size_t c = 0;
        while (c < classes.size() && !same_class(devices[classes[c].devs[0]].input, devices[d].input)) c++;
        if (c == classes.size()) classes.push_back(hip_class());
        classes[c].devs.push_back(d);
    }
    for (size_t c = 0; c < classes.size(); c++) prepare_class(classes[c], hip_device);

    while (!do_exit) {
        if (devices_running == 0) {  // rtl_airband.cpp:377-381
            log(LOG_ERR, "All receivers failed, exiting\n");
            do_exit = 1;
            continue;
        }
        bool worked = false;
        for (size_t c = 0; c < classes.size(); c++) {
            hip_class& k = classes[c];
            const int n = (int)k.devs.size();
            // 1. per device: a failed input is taken out exactly as demodulate() does it (rtl_airband.cpp:383-391) -- and out of the
            //    handle, so that the others are not held up waiting for its bytes; a running one hands over what its rx thread
            //    appended (circbuffer_append, input-helpers.cpp:37-63).  Cursor discipline of rtl_airband.cpp:370-375 and :669: bufe
            //    is read under buffer_lock, bufs is ours.
            for (int i = 0; i < n; i++) {
                device_t* dev = devices + k.devs[i];
                input_t* in = dev->input;
                if (in->state != INPUT_RUNNING) {
                    if (in->state == INPUT_FAILED) {
                        in->state = INPUT_DISABLED;
                        disable_device_outputs(dev);
                        devices_running--;
                        airband_hip_device_enable(k.h, i, 0);
                    }
                    continue;
                }
                pthread_mutex_lock(&in->buffer_lock);
                const size_t bufe = in->bufe;
                pthread_mutex_unlock(&in->buffer_lock);
                while (in->bufs != bufe) {
                    const size_t run = (bufe > in->bufs ? bufe : in->buf_size) - in->bufs;  // contiguous part of the ring
                    const int64_t took = airband_hip_submit(k.h, i, in->buffer + in->bufs, run);
                    if (took <= 0) break;  // staging full: the GPU is behind, try again next round
                    in->bufs = (in->bufs + (size_t)took) % in->buf_size;
                }
            }
            // 2. one WAVE_BATCH for every running device of the class, once all of them have the bytes (availability rule :394-400)
            const int rc = airband_hip_process(k.h);
            if (rc == AIRBAND_HIP_EAGAIN) continue;
            if (rc < 0) {
                log(LOG_CRIT, "airband_hip: %s\n", airband_hip_last_error(k.h));
                error();
            }
            if (airband_hip_collect(k.h, k.wave.data(), k.iq.data(), k.axc.data(), k.st.data()) != AIRBAND_HIP_OK) continue;  // nothing to publish
            worked = true;
            // 3. publish what the per-channel loop publishes (rtl_airband.cpp:549-619,:645-655); the library has already done the
            //    consumer's tail copy (output.cpp:920), so the samples go where process_outputs() reads them
            size_t q = 0;
            for (int i = 0; i < n; i++) {
                device_t* dev = devices + k.devs[i];
                if (dev->input->state != INPUT_RUNNING) {  // taken out above: its channels keep what they last held
                    q += dev->channel_count;
                    continue;
                }
                for (int j = 0; j < dev->channel_count; j++, q++) {
                    channel_t* ch = dev->channels + j;
                    memcpy(ch->waveout, &k.wave[q * k.g.wave_batch], sizeof(float) * k.g.wave_batch);
                    if (ch->has_iq_outputs) memcpy(ch->iq_out, &k.iq[q * k.g.wave_batch * 2], sizeof(float) * 2 * k.g.wave_batch);
                    ch->axcindicate = (status)k.axc[q];
                    ch->freqlist->active_counter = k.st[q].active_counter;
                    ch->freqlist->agcavgfast = k.st[q].agcavgfast;
                    ch->freqlist->squelch.mirror(k.st[q].noise_level, k.st[q].signal_level, k.st[q].squelch_level, k.st[q].open_count, k.st[q].flappy_count,
                                                 k.st[q].ctcss_count, k.st[q].no_ctcss_count);
                }
                if (dev->waveavail == 1) {  // rtl_airband.cpp:649-654: the output thread has not drained the previous batch
                    dev->output_overrun_count++;
                } else {
                    dev->waveavail = 1;
                }
            }
            dp->mp3_signal->send();  // rtl_airband.cpp:662
        }
        if (!worked) SLEEP(1);
    }
    for (size_t c = 0; c < classes.size(); c++) airband_hip_release(classes[c].h);  // like gpu_fft_release on do_exit (rtl_airband.cpp:360-365)
    return NULL;
}
