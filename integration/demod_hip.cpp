// demod_hip.cpp -- RTLSDR-Airband side of the MI355X backend: demodulate() for a shard of devices, done by
// libairband_hip.so (C ABI: airband_hip.h).  Added to the reference tree by integration/airband_hip.patch together with
// the few fields / setters it uses; everything else -- main(), the config parser, the input drivers and their ring
// buffers, output threads, the stats file -- is untouched.
//
// Selected like the VideoCore FFT backend (WITH_BCM_VC, rtl_airband.cpp:293-310): a compile-time switch, here
// WITH_AIRBAND_HIP, which makes main() start demodulate_hip() instead of demodulate() (rtl_airband.cpp:1110-1112).
//
// A shard (device_start .. device_end of demod_params_t) is spread over the GPUs of the node: its devices are grouped into
// classes (sample format, hop), every class is cut into contiguous ranges, one per GPU, and each range is one library handle
// ("part").  Mixers all of whose inputs lie in one class are summed on the GPUs: every part reduces its own inputs, the partial
// sums meet over RCCL / xGMI (airband_hip_allreduce_mixers), and the result is published where mixer_thread() would have put it.
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <libconfig.h++>
#include <vector>

#include "airband_hip.h"
#include "rtl_airband.h"

extern int tui;  // rtl_airband.cpp:75 (set by -f / cleared by -F; only rtl_airband.cpp reads it in the reference)

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

// mixer_thread() (mixer.cpp:157) calls this with the Signal of the output thread that serves the mixers (rtl_airband.cpp:1097-1100)
static Signal* volatile g_mixer_signal = NULL;
void demod_hip_mixer_signal(Signal* signal) {
    g_mixer_signal = signal;
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

// ---- which GPUs --------------------------------------------------------------------------------------------------------
// HIP device indices the backend may use: all of them, or the list in AIRBAND_HIP_GPUS ("0,1,2,3"; an index may repeat --
// "0,0" runs two parts on one GPU, which is how the partition is tested on a one-GPU box).
static std::vector<int> gpu_list() {
    std::vector<int> out;
    const int have = airband_hip_gpu_count();
    const char* env = getenv("AIRBAND_HIP_GPUS");
    if (env && *env) {
        for (const char* p = env; *p;) {
            char* end = NULL;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            if (v >= 0 && (have == 0 || v < have)) out.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
    }
    if (out.empty())
        for (int g = 0; g < (have > 0 ? have : 1); g++) out.push_back(g);
    return out;
}

// Contiguous ranges of a class's devices over n_gpus GPUs: GPU g takes [first[g], first[g + 1]) -- the partition the reference itself
// uses for its demodulator threads (rtl_airband.cpp:1052-1086: contiguous device_start / device_end).  Plain arithmetic, exported for the tests.
extern "C" void demod_hip_partition(int n_devices, int n_gpus, int* first) {
    if (n_gpus < 1) n_gpus = 1;
    for (int g = 0; g <= n_gpus; g++) first[g] = (int)((long)n_devices * g / n_gpus);
}

// ---- parts, classes ----------------------------------------------------------------------------------------------------
// One library handle per PART: the devices of one class (libairband_hip batches the dongles of a handle through one launch, so they
// must share sample format and hop, round(sample_rate / WAVE_RATE), rtl_airband.cpp:394) that one GPU demodulates.
struct hip_part {
    airband_hip_handle* h;
    int gpu;
    std::vector<int> devs;     // indices into devices[]
    std::vector<char> parked;  // per device: switched off in the handle because its input is not INPUT_RUNNING
    int live;                  // devices that are neither failed nor parked
    airband_hip_geometry g;
    std::vector<float> wave, iq;
    std::vector<char> axc;
    std::vector<airband_hip_channel_stats> st;
    bool have_batch;
};

struct hip_class {
    std::vector<int> devs;
    std::vector<hip_part> parts;
    std::vector<int> leader;         // per part: the first part on the same GPU (partial sums of one GPU are added up there)
    std::vector<int> fabric;         // the leaders: one per distinct GPU, all-reduced over RCCL when there are several
    std::vector<int> served;         // indices into mixers[]: the mixers this class sums on the GPUs
    std::vector<float> mix_left, mix_right;
    std::vector<uint8_t> mix_signal;
};

static bool same_class(const input_t* a, const input_t* b) {
    return a->sfmt == b->sfmt && a->bytes_per_sample == b->bytes_per_sample &&
           round((double)a->sample_rate / (double)WAVE_RATE) == round((double)b->sample_rate / (double)WAVE_RATE);
}

static void prepare_part(hip_part& k) {
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
    cfg.hip_device = k.gpu;
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
    k.parked.assign(n, 0);
    k.live = n;
    k.have_batch = false;
}

static void hip_check(airband_hip_handle* h, int rc, const char* what) {
    if (rc == AIRBAND_HIP_OK) return;
    log(LOG_CRIT, "airband_hip: %s: %s\n", what, airband_hip_last_error(h));
    error();
}

// Mixers whose inputs ALL come from channels of this class are served on the GPUs (mixer_connect_input(), mixer.cpp:57-94, has
// recorded ampfactor / ampl / ampr per input; the connection order inside a mixer is its input index, which is also the order
// mix_waveforms() adds them in, mixer.cpp:189-214).  Any other mixer stays with mixer_thread().
struct conn { int part, dev_in_part, chan, input; };  // one O_MIXER output of a channel: where the channel lives, which input of the mixer it feeds
static void wire_mixers(hip_class& c, int device_start, int device_end, bool mark_only) {
    std::vector<std::vector<conn> > per_mixer(mixer_count);
    std::vector<char> foreign(mixer_count, 0);
    std::vector<int> part_of(device_count, -1), idx_in_part(device_count, -1);
    for (size_t p = 0; p < c.parts.size(); p++)
        for (size_t i = 0; i < c.parts[p].devs.size(); i++) {
            part_of[c.parts[p].devs[i]] = (int)p;
            idx_in_part[c.parts[p].devs[i]] = (int)i;
        }
    if (mark_only)  // the parts do not exist yet: which mixers are served depends on the class alone
        for (size_t i = 0; i < c.devs.size(); i++) part_of[c.devs[i]] = 0;
    for (int d = 0; d < device_count; d++) {
        const bool ours = d >= device_start && d < device_end && part_of[d] >= 0;
        for (int j = 0; j < devices[d].channel_count; j++) {
            channel_t* ch = devices[d].channels + j;
            for (int o = 0; o < ch->output_count; o++) {
                if (ch->outputs[o].type != O_MIXER) continue;
                mixer_data* md = (mixer_data*)ch->outputs[o].data;
                const int m = (int)(md->mixer - mixers);
                if (m < 0 || m >= mixer_count) continue;
                if (!ours) {
                    foreign[m] = 1;
                    continue;
                }
                conn k = {part_of[d], idx_in_part[d], j, md->input};
                per_mixer[m].push_back(k);
            }
        }
    }
    if (mark_only) {
        // first thing, before the HIP runtime comes up and the handles are built (which takes a while): from here on mixer_thread() and
        // mixer_put_samples() leave these mixers alone -- a mixer_thread() that still saw them would emit an (empty) batch every third interval
        // (mixer.cpp:225-248)
        for (int m = 0; m < mixer_count; m++) {
            if (!mixers[m].enabled || foreign[m] || per_mixer[m].empty() || (int)per_mixer[m].size() != mixers[m].input_count) continue;
            c.served.push_back(m);
            mixers[m].gpu_served = true;
        }
        return;
    }
    if (c.served.empty()) return;
    const int S = (int)c.served.size();
    for (size_t p = 0; p < c.parts.size(); p++) {
        std::vector<airband_hip_mixer_input> in;
        for (int s = 0; s < S; s++) {
            const mixer_t* mx = mixers + c.served[s];
            for (int input = 0; input < mx->input_count; input++)  // connection order = input index
                for (size_t q = 0; q < per_mixer[c.served[s]].size(); q++) {
                    const conn& k = per_mixer[c.served[s]][q];
                    if (k.input != input || k.part != (int)p) continue;
                    const mixinput_t* mi = mx->inputs + input;
                    airband_hip_mixer_input e;
                    e.device = k.dev_in_part;
                    e.channel = k.chan;
                    e.mixer = s;
                    e.ampfactor = mi->ampfactor;
                    // ampl = min(1, 1 - balance), ampr = min(1, 1 + balance) (mixer.cpp:82-83): one of them is 1
                    e.balance = mi->ampl < 1.0f ? 1.0f - mi->ampl : (mi->ampr < 1.0f ? mi->ampr - 1.0f : 0.0f);
                    in.push_back(e);
                }
        }
        hip_check(c.parts[p].h, airband_hip_set_mixers(c.parts[p].h, S, in.empty() ? NULL : in.data(), (int32_t)in.size()), "set_mixers");
        for (int s = 0; s < S; s++)
            if (mixers[c.served[s]].channel.mode == MM_STEREO) hip_check(c.parts[p].h, airband_hip_mixer_set_stereo(c.parts[p].h, s, 1), "mixer_set_stereo");
    }
    c.mix_left.resize((size_t)S * WAVE_BATCH);
    c.mix_right.resize((size_t)S * WAVE_BATCH);
    c.mix_signal.resize(S);
    // the parts of one GPU add their sums up on that GPU; the GPUs exchange theirs over RCCL
    // (AIRBAND_HIP_FABRIC_PER_PART=1: every part is a fabric rank of its own even where parts share a GPU -- RCCL proper refuses that; the test suite
    // sets it together with AIRBAND_HIP_RCCL_LIB=<its in-process stand-in> to run the exchange below with several ranks on a one-GPU box)
    const char* per_part = getenv("AIRBAND_HIP_FABRIC_PER_PART");
    c.leader.assign(c.parts.size(), 0);
    for (size_t p = 0; p < c.parts.size(); p++) {
        size_t l = 0;
        while (c.parts[l].gpu != c.parts[p].gpu) l++;
        if (per_part && *per_part == '1') l = p;
        c.leader[p] = (int)l;
        if (l == p) c.fabric.push_back((int)p);
    }
    if (c.fabric.size() > 1) {
        std::vector<airband_hip_handle*> hs;
        for (size_t i = 0; i < c.fabric.size(); i++) hs.push_back(c.parts[c.fabric[i]].h);
        hip_check(hs[0], airband_hip_comm_init_all(hs.data(), (int32_t)hs.size()), "comm_init_all");
    }
}

// ---- a few threads for the per-part work (one per part beyond the first: copying into pinned rings and out of result buffers is what
// the host does here, and one core does not feed eight GPUs) ---------------------------------------------------------------------------
struct hip_pool {
    pthread_mutex_t lock;
    pthread_cond_t go, done;
    std::vector<pthread_t> threads;
    void (*fn)(hip_class*, int);
    hip_class* cls;
    int n_tasks, next, pending;
    unsigned long generation;
    bool quit;
};
static void* pool_main(void* arg) {
    hip_pool* P = (hip_pool*)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&P->lock);
    for (;;) {
        while (!P->quit && (P->generation == seen || P->next >= P->n_tasks)) {
            if (P->generation != seen && P->next >= P->n_tasks) seen = P->generation;
            pthread_cond_wait(&P->go, &P->lock);
        }
        if (P->quit) break;
        const int t = P->next++;
        pthread_mutex_unlock(&P->lock);
        P->fn(P->cls, t);
        pthread_mutex_lock(&P->lock);
        if (--P->pending == 0) pthread_cond_signal(&P->done);
    }
    pthread_mutex_unlock(&P->lock);
    return NULL;
}
static void pool_start(hip_pool& P, int n_threads) {
    pthread_mutex_init(&P.lock, NULL);
    pthread_cond_init(&P.go, NULL);
    pthread_cond_init(&P.done, NULL);
    P.fn = NULL;
    P.cls = NULL;
    P.n_tasks = P.next = P.pending = 0;
    P.generation = 0;
    P.quit = false;
    P.threads.resize(n_threads > 0 ? n_threads : 0);
    for (size_t i = 0; i < P.threads.size(); i++) pthread_create(&P.threads[i], NULL, pool_main, &P);
}
// fn(cls, part) for every part of the class: part 0 on the calling thread, the others on the pool
static void pool_run(hip_pool& P, hip_class* cls, void (*fn)(hip_class*, int)) {
    const int n = (int)cls->parts.size();
    if (n <= 1 || P.threads.empty()) {
        for (int t = 0; t < n; t++) fn(cls, t);
        return;
    }
    pthread_mutex_lock(&P.lock);
    P.fn = fn;
    P.cls = cls;
    P.n_tasks = n;
    P.next = 1;
    P.pending = n - 1;
    P.generation++;
    pthread_cond_broadcast(&P.go);
    pthread_mutex_unlock(&P.lock);
    fn(cls, 0);
    pthread_mutex_lock(&P.lock);
    while (P.pending > 0) pthread_cond_wait(&P.done, &P.lock);
    pthread_mutex_unlock(&P.lock);
}
static void pool_stop(hip_pool& P) {
    pthread_mutex_lock(&P.lock);
    P.quit = true;
    pthread_cond_broadcast(&P.go);
    pthread_mutex_unlock(&P.lock);
    for (size_t i = 0; i < P.threads.size(); i++) pthread_join(P.threads[i], NULL);
}

// A running input hands over what its rx thread appended (circbuffer_append, input-helpers.cpp:37-63).  Cursor discipline of
// rtl_airband.cpp:370-375 and :669: bufe is read under buffer_lock, bufs is ours.
static void part_feed(hip_class* c, int p) {
    hip_part& k = c->parts[p];
    for (size_t i = 0; i < k.devs.size(); i++) {
        input_t* in = devices[k.devs[i]].input;
        if (in->state != INPUT_RUNNING || k.parked[i]) continue;
        pthread_mutex_lock(&in->buffer_lock);
        const size_t bufe = in->bufe;
        pthread_mutex_unlock(&in->buffer_lock);
        while (in->bufs != bufe) {
            const size_t run = (bufe > in->bufs ? bufe : in->buf_size) - in->bufs;  // contiguous part of the ring
            const int64_t took = airband_hip_submit(k.h, (int32_t)i, in->buffer + in->bufs, run);
            if (took <= 0) break;  // staging full: the GPU is behind, try again next round
            in->bufs = (in->bufs + (size_t)took) % in->buf_size;
        }
    }
}

// Results of a part, and what the per-channel loop publishes (rtl_airband.cpp:549-619,:645-655); the library has already done the
// consumer's tail copy (output.cpp:920), so the samples go where process_outputs() reads them.
static void part_collect(hip_class* c, int p) {
    hip_part& k = c->parts[p];
    if (!k.have_batch) return;
    if (airband_hip_collect(k.h, k.wave.data(), k.iq.data(), k.axc.data(), k.st.data()) != AIRBAND_HIP_OK) {
        k.have_batch = false;  // nothing to publish
        return;
    }
    size_t q = 0;
    for (size_t i = 0; i < k.devs.size(); i++) {
        device_t* dev = devices + k.devs[i];
        if (dev->input->state != INPUT_RUNNING || k.parked[i]) {  // taken out: its channels keep what they last held
            q += dev->channel_count;
            continue;
        }
        for (int j = 0; j < dev->channel_count; j++, q++) {
            channel_t* ch = dev->channels + j;
            const airband_hip_channel_stats& st = k.st[q];
            memcpy(ch->waveout, &k.wave[q * k.g.wave_batch], sizeof(float) * k.g.wave_batch);
            if (ch->has_iq_outputs) memcpy(ch->iq_out, &k.iq[q * k.g.wave_batch * 2], sizeof(float) * 2 * k.g.wave_batch);
            ch->axcindicate = (status)k.axc[q];
            ch->freqlist->active_counter = st.active_counter;
            ch->freqlist->agcavgfast = st.agcavgfast;
            ch->freqlist->squelch.mirror(st.noise_level, st.signal_level, st.squelch_level, st.open_count, st.flappy_count, st.ctcss_count, st.no_ctcss_count,
                                         st.signal_outside_filter != 0);
        }
    }
}

// The waterfall line of one device for the batch just published (rtl_airband.cpp:632-643) and its scroll (:663-667)
static void tui_line(int device_num, device_t* dev) {
    for (int i = 0; i < dev->channel_count; i++) {
        channel_t* channel = dev->channels + i;
        freq_t* fparms = channel->freqlist;
        char symbol = fparms->squelch.signal_outside_filter() ? '~' : (char)channel->axcindicate;
        GOTOXY(i * 10, device_num * 17 + dev->row + 3);
        printf("%4.0f/%3.0f%c ", level_to_dBFS(fparms->squelch.signal_level()), level_to_dBFS(fparms->squelch.noise_level()), symbol);
        fflush(stdout);
    }
}

void* demodulate_hip(void* params) {
    demod_params_t* dp = (demod_params_t*)params;  // device_start / device_end shard (rtl_airband.h demod_params_t)
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
    // A class goes over the GPUs in contiguous ranges.  With multiple_demod_threads a shard is ONE device (rtl_airband.cpp:1052-1086):
    // the shards then go round robin, device_start picks the GPU.
    for (size_t c = 0; c < classes.size(); c++) wire_mixers(classes[c], dp->device_start, dp->device_end, true);
    const std::vector<int> gpus = gpu_list();  // the first call into the HIP runtime
    size_t max_parts = 1;
    for (size_t c = 0; c < classes.size(); c++) {
        hip_class& cls = classes[c];
        const int n = (int)cls.devs.size();
        const int G = (int)gpus.size() < n ? (int)gpus.size() : n;
        std::vector<int> first(G + 1);
        demod_hip_partition(n, G, first.data());
        for (int g = 0; g < G; g++) {
            if (first[g + 1] == first[g]) continue;
            hip_part part;
            part.h = NULL;
            part.gpu = gpus[(dp->device_start + g) % gpus.size()];
            part.devs.assign(cls.devs.begin() + first[g], cls.devs.begin() + first[g + 1]);
            cls.parts.push_back(part);
        }
        for (size_t p = 0; p < cls.parts.size(); p++) prepare_part(cls.parts[p]);
        wire_mixers(cls, dp->device_start, dp->device_end, false);
        if (cls.parts.size() > max_parts) max_parts = cls.parts.size();
    }
    hip_pool pool;
    pool_start(pool, (int)max_parts - 1);

    while (!do_exit) {
        if (devices_running == 0) {  // rtl_airband.cpp:377-381
            log(LOG_ERR, "All receivers failed, exiting\n");
            do_exit = 1;
            continue;
        }
        bool worked = false;
        for (size_t c = 0; c < classes.size(); c++) {
            hip_class& cls = classes[c];
            // 1. per device: a failed input is taken out exactly as demodulate() does it (rtl_airband.cpp:383-391) -- and out of its handle, so
            //    that the others are not held up waiting for its bytes.  Any other state than INPUT_RUNNING is passed by, as demodulate() does:
            //    the device is parked in the handle (no bytes are expected from it) and taken back when it runs again.
            for (size_t p = 0; p < cls.parts.size(); p++) {
                hip_part& k = cls.parts[p];
                for (size_t i = 0; i < k.devs.size(); i++) {
                    device_t* dev = devices + k.devs[i];
                    input_t* in = dev->input;
                    if (in->state == INPUT_FAILED) {
                        in->state = INPUT_DISABLED;
                        disable_device_outputs(dev);
                        devices_running--;
                        if (!k.parked[i]) {
                            airband_hip_device_enable(k.h, (int32_t)i, 0);
                            k.live--;
                        }
                        k.parked[i] = 2;  // for good
                    } else if (in->state != INPUT_RUNNING && !k.parked[i]) {
                        airband_hip_device_enable(k.h, (int32_t)i, 0);
                        k.parked[i] = 1;
                        k.live--;
                    } else if (in->state == INPUT_RUNNING && k.parked[i] == 1) {
                        // back: it joins the others at their stream position; what its ring still holds is older than that
                        pthread_mutex_lock(&in->buffer_lock);
                        in->bufs = in->bufe;
                        pthread_mutex_unlock(&in->buffer_lock);
                        airband_hip_device_enable(k.h, (int32_t)i, 1);
                        k.parked[i] = 0;
                        k.live++;
                    }
                }
            }
            pool_run(pool, &cls, part_feed);
            // 2. one WAVE_BATCH for every running device of a part, once all of them have the bytes (availability rule :394-400).  Parts whose
            //    mixer sums meet advance together: every one of them must be ready.
            bool lockstep_ready = true;
            if (!cls.served.empty())
                for (size_t p = 0; p < cls.parts.size(); p++)
                    if (cls.parts[p].live > 0 && airband_hip_batch_ready(cls.parts[p].h) != AIRBAND_HIP_OK) lockstep_ready = false;
            if (!lockstep_ready) continue;
            bool any = false;
            for (size_t p = 0; p < cls.parts.size(); p++) {
                hip_part& k = cls.parts[p];
                k.have_batch = false;
                if (k.live == 0) continue;
                const int rc = airband_hip_process(k.h);
                if (rc == AIRBAND_HIP_EAGAIN) continue;
                if (rc < 0) {
                    log(LOG_CRIT, "airband_hip: %s\n", airband_hip_last_error(k.h));
                    error();
                }
                k.have_batch = true;
                any = true;
            }
            if (!any) continue;
            // 3. the mixer exchange (mixer.cpp:133-140,201-214), enqueued on the GPUs behind the batches: the parts of one GPU add up on
            //    its first part, the GPUs all-reduce over RCCL.  A part without a batch (all its devices gone: demodulate() has called
            //    mixer_disable_input() for every one of their outputs, mixer.cpp:96-112, rtl_airband.cpp:383-391, and keeps mixing the rest)
            //    adds nothing -- its buffers have to be CLEARED for that: they hold its last batch's sums, a leader's hold what the other parts
            //    of its GPU added, a fabric rank's the whole node's (the all-reduce is in place).
            if (!cls.served.empty()) {
                for (size_t p = 0; p < cls.parts.size(); p++)
                    if (!cls.parts[p].have_batch) hip_check(cls.parts[p].h, airband_hip_clear_mixers(cls.parts[p].h), "clear_mixers");
                for (size_t p = 0; p < cls.parts.size(); p++)
                    if (cls.leader[p] != (int)p && cls.parts[p].have_batch)
                        hip_check(cls.parts[cls.leader[p]].h, airband_hip_add_mixers(cls.parts[cls.leader[p]].h, cls.parts[p].h), "add_mixers");
                if (cls.fabric.size() > 1) {
                    hip_check(NULL, airband_hip_comm_group_begin(), "comm_group_begin");
                    for (size_t i = 0; i < cls.fabric.size(); i++)
                        hip_check(cls.parts[cls.fabric[i]].h, airband_hip_allreduce_mixers(cls.parts[cls.fabric[i]].h, NULL), "allreduce_mixers");
                    hip_check(NULL, airband_hip_comm_group_end(), "comm_group_end");
                }
            }
            pool_run(pool, &cls, part_collect);
            // 4. hand-off per device (rtl_airband.cpp:645-667), waterfall included
            for (size_t p = 0; p < cls.parts.size(); p++) {
                hip_part& k = cls.parts[p];
                if (!k.have_batch) continue;
                worked = true;
                for (size_t i = 0; i < k.devs.size(); i++) {
                    device_t* dev = devices + k.devs[i];
                    if (dev->input->state != INPUT_RUNNING || k.parked[i]) continue;
                    if (tui) tui_line(k.devs[i], dev);
                    if (dev->waveavail == 1) {  // rtl_airband.cpp:649-654: the output thread has not drained the previous batch
                        dev->output_overrun_count++;
                    } else {
                        dev->waveavail = 1;
                    }
                    dev->row++;
                    if (dev->row == 12) dev->row = 0;
                }
            }
            dp->mp3_signal->send();  // rtl_airband.cpp:662
            // 5. the mixers this class serves: what mixer_thread() leaves in mixer->channel (mixer.cpp:189-248).  Part 0 leads its GPU and is a
            //    fabric rank, so after the exchange it holds the class's sums -- whether or not it ran a batch itself (`any` says some part did)
            if (!cls.served.empty() &&
                airband_hip_collect_mixers(cls.parts[0].h, cls.mix_left.data(), cls.mix_right.data(), cls.mix_signal.data()) == AIRBAND_HIP_OK) {
                bool sent = false;
                for (size_t s = 0; s < cls.served.size(); s++) {
                    mixer_t* mixer = mixers + cls.served[s];
                    if (!mixer->enabled) continue;
                    channel_t* channel = &mixer->channel;
                    if (channel->state == CH_READY) mixer->output_overrun_count++;  // previous output not yet handled by the output thread (mixer.cpp:180-187)
                    memcpy(channel->waveout, &cls.mix_left[s * WAVE_BATCH], WAVE_BATCH * sizeof(float));
                    if (channel->mode == MM_STEREO) memcpy(channel->waveout_r, &cls.mix_right[s * WAVE_BATCH], WAVE_BATCH * sizeof(float));
                    channel->axcindicate = cls.mix_signal[s] ? SIGNAL : NO_SIGNAL;
                    channel->state = CH_READY;
                    sent = true;
                }
                Signal* ms = g_mixer_signal ? g_mixer_signal : dp->mp3_signal;  // mixer_thread() announces the mixers' output thread when it starts
                if (sent) ms->send();
            }
        }
        if (!worked) SLEEP(1);
    }
    pool_stop(pool);
    for (size_t c = 0; c < classes.size(); c++)
        for (size_t p = 0; p < classes[c].parts.size(); p++) airband_hip_release(classes[c].parts[p].h);  // like gpu_fft_release on do_exit (rtl_airband.cpp:360-365)
    return NULL;
}
