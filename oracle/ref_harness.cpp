/* oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Builds the REAL reference hot path into oracle/_ref/: this translation unit #includes the
 * reference's src/rtl_airband.cpp where it lies (so `static int devices_running`, demodulate(), AFC,
 * the FM helpers etc. are the reference's own object code) and adds a small C API around it.
 * No reference source is copied into this repository.
 *
 * What this file restates (because the reference only does it inside the libconfig++ parser,
 * which cannot run here) is the hand-construction of device_t / channel_t / freq_t / input_t:
 *   mk_freqlist()            src/config.cpp:265-281
 *   channel defaults         src/config.cpp:313-331
 *   squelch thresholds       src/config.cpp:436-515
 *   notch / ctcss / bandwidth / ampfactor / tau   src/config.cpp:516-650
 *   bins                     src/config.cpp:666-667
 *   dm_dphi                  src/config.cpp:679-712
 *   ring sizing              src/config.cpp:796-806
 *   file-input defaults      src/input-file.cpp:162-181
 * and the roles of the two neighbouring threads:
 *   rx thread  -> circbuffer_append()          src/input-helpers.cpp:37-63 (the reference's own function)
 *   output thread -> read waveout[0..WAVE_BATCH), tail copy, clear waveavail   src/output.cpp:917-922
 */
#define main reference_main
#include "rtl_airband.cpp" /* found through -I/root/reference/src */
#undef main

#include <dlfcn.h>
#include <limits.h>
#include <sched.h>

#include <vector>

#include "../include/airband_hip.h"
#include "input-file.h" /* file_dev_data_t: the reference's file input (src/input-file.cpp) is compiled in */
MODULE_EXPORT input_t* file_input_new(); /* src/input-file.cpp:162-181; input_new() finds it with dlsym in the real binary (src/input-common.cpp:35-54) */
#include "input-helpers.h"

/* ---- symbols main()/demodulate() reference but the oracle never reaches ---------------------- */
char const* RTL_AIRBAND_VERSION = "oracle-harness";
static void refh_unreachable(const char* what) {
    fprintf(stderr, "oracle/ref_harness: %s must never be reached\n", what);
    abort();
}
int parse_devices(libconfig::Setting&) { refh_unreachable("parse_devices"); return 0; }
int parse_mixers(libconfig::Setting&) { refh_unreachable("parse_mixers"); return 0; }
lame_t airlame_init(mix_modes, int, int) { refh_unreachable("airlame_init"); return NULL; }
void shout_setup(icecast_data*, mix_modes) { refh_unreachable("shout_setup"); }
static int g_outputs_disabled[4096]; /* disable_device_outputs() calls per device (src/output.cpp:591-596 is the real one) */
void disable_device_outputs(device_t* dev) {
    const long d = dev - devices;
    if (d >= 0 && d < 4096) g_outputs_disabled[d]++;
}
void disable_channel_outputs(channel_t*) {}
void* output_check_thread(void*) { refh_unreachable("output_check_thread"); return NULL; }
void* output_thread(void*) { refh_unreachable("output_thread"); return NULL; }
/* (mixer_thread, getmixerbyname, mixer_get_error: the reference's own src/mixer.cpp is compiled in) */

/* ---- harness state --------------------------------------------------------------------------- */
#define REFH_MAX_THREADS 256
static pthread_t g_demod_thread[REFH_MAX_THREADS];
static int g_threads_running = 0;
static demod_params_t g_demod_params[REFH_MAX_THREADS];
static Signal g_signal;
static size_t* g_hops_fed_bytes = NULL;  /* total bytes fed per device */
static char g_trace_dir[512] = "";

extern "C" {

int refh_wave_rate(void) { return WAVE_RATE; }
int refh_wave_batch(void) { return WAVE_BATCH; }
int refh_agc_extra(void) { return AGC_EXTRA; }
int refh_has_nfm(void) {
#ifdef NFM
    return 1;
#else
    return 0;
#endif
}

void refh_set_trace_dir(const char* dir) { snprintf(g_trace_dir, sizeof(g_trace_dir), "%s", dir ? dir : ""); }

int refh_init(int n_devices, int fft_log, int fm_demod_algo, int global_tau_us) {
    if (fft_log < MIN_FFT_SIZE_LOG || fft_log > MAX_FFT_SIZE_LOG) return -2;
    log_destination = NONE;
    fft_size_log = (size_t)fft_log;
    fft_size = (size_t)1 << fft_size_log;
    device_count = n_devices;
    devices = (device_t*)XCALLOC(n_devices, sizeof(device_t));
    g_hops_fed_bytes = (size_t*)XCALLOC(n_devices, sizeof(size_t));
    mixer_count = 0;
    do_exit = 0;
#ifdef NFM
    fm_demod = fm_demod_algo ? FM_QUADRI_DEMOD : FM_FAST_ATAN2;
    /* global tau: src/rtl_airband.cpp:87 default, :825-826 override */
    alpha = exp(-1.0f / (WAVE_RATE * 2e-4));
    if (global_tau_us >= 0) alpha = (global_tau_us == 0 ? 0.0f : exp(-1.0f / (WAVE_RATE * 1e-6 * global_tau_us)));
#else
    (void)fm_demod_algo;
    (void)global_tau_us;
#endif
    sincosf_lut_init(); /* src/rtl_airband.cpp:1107 */
    return 0;
}

int refh_add_device(int d, const airband_hip_device_cfg* cfg) {
    device_t* dev = devices + d;
#ifdef WITH_AIRBAND_HIP
    dev->cfg_tau = cfg->tau_us; /* what the patched parse_devices() keeps (integration/airband_hip.patch, src/config.cpp:776) */
#endif
    input_t* input = (input_t*)XCALLOC(1, sizeof(input_t));
    input->state = INPUT_RUNNING;
    input->sfmt = (sample_format_t)cfg->sfmt;
    switch (cfg->sfmt) { /* driver defaults: src/input-file.cpp:171-173, src/input-soapysdr.cpp:45-64 */
        case SFMT_U8:
        case SFMT_S8:
            input->bytes_per_sample = 1;
            input->fullscale = (float)SCHAR_MAX - 0.5f;
            break;
        case SFMT_S16:
            input->bytes_per_sample = 2;
            input->fullscale = (float)SHRT_MAX - 0.5f;
            break;
        case SFMT_F32:
            input->bytes_per_sample = 4;
            input->fullscale = 1.0f;
            break;
        default:
            return -4;
    }
    if (cfg->fullscale > 0) input->fullscale = cfg->fullscale;
    input->sample_rate = cfg->sample_rate;
    input->centerfreq = cfg->centerfreq;
    pthread_mutex_init(&input->buffer_lock, NULL);
    dev->input = input;
    dev->mode = R_MULTICHANNEL;
#ifdef NFM
    dev->alpha = alpha; /* src/config.cpp:774-778 */
    if (cfg->tau_us >= 0) dev->alpha = (cfg->tau_us == 0 ? 0.0f : exp(-1.0f / (WAVE_RATE * 1e-6 * cfg->tau_us)));
#endif
    /* ring sizing: src/config.cpp:796-806 */
    size_t fft_batch_len = FFT_BATCH * (2 * input->bytes_per_sample * (size_t)ceil((double)input->sample_rate / (double)WAVE_RATE));
    input->buf_size = MIN_BUF_SIZE;
    if (input->buf_size % fft_batch_len != 0) input->buf_size += fft_batch_len - input->buf_size % fft_batch_len;
    input->buffer = (unsigned char*)XCALLOC(sizeof(unsigned char), input->buf_size + 2 * input->bytes_per_sample * fft_size);
    input->bufs = input->bufe = 0;
    input->overflow_count = 0;
    dev->output_overrun_count = 0;
    dev->waveend = dev->waveavail = dev->row = dev->tq_head = dev->tq_tail = 0;
    dev->last_frequency = -1;

    dev->channel_count = cfg->channel_count;
    dev->channels = (channel_t*)XCALLOC(cfg->channel_count, sizeof(channel_t));
    dev->bins = (size_t*)XCALLOC(cfg->channel_count, sizeof(size_t));
    dev->base_bins = (size_t*)XCALLOC(cfg->channel_count, sizeof(size_t));

    for (int j = 0; j < cfg->channel_count; j++) {
        const airband_hip_channel_cfg* cc = cfg->channels + j;
        channel_t* channel = dev->channels + j;
        for (int k = 0; k < AGC_EXTRA; k++) { /* src/config.cpp:313-316 */
            channel->wavein[k] = 20;
            channel->waveout[k] = 0.5;
        }
        channel->axcindicate = NO_SIGNAL;
        channel->mode = MM_MONO;
        channel->freq_count = 1;
        channel->freq_idx = 0;
        channel->highpass = 100;
        channel->lowpass = 2500;
#ifdef NFM
        channel->pr = 0;
        channel->pj = 0;
        channel->prev_waveout = 0.5;
        channel->alpha = dev->alpha;
#endif
        channel->afc = (unsigned char)cc->afc;
        /* mk_freqlist(1): src/config.cpp:265-281 */
        freq_t* fl = (freq_t*)XCALLOC(1, sizeof(freq_t));
        fl[0].frequency = cc->frequency;
        fl[0].label = NULL;
        fl[0].agcavgfast = 0.5f;
        fl[0].ampfactor = 1.0f;
        fl[0].squelch = Squelch();
        fl[0].active_counter = 0;
        fl[0].modulation = MOD_AM;
#ifdef NFM
        if (cc->modulation == AIRBAND_MOD_NFM) fl[0].modulation = MOD_NFM;
#else
        if (cc->modulation != AIRBAND_MOD_AM) return -4;
#endif
        channel->freqlist = fl;
        /* squelch thresholds: src/config.cpp:436-515 */
        if (cc->squelch_threshold_dbfs < 0) {
            fl[0].squelch.set_squelch_level_threshold(dBFS_to_level((float)cc->squelch_threshold_dbfs));
        } else {
            fl[0].squelch.set_squelch_level_threshold(0);
        }
        if (cc->squelch_snr_threshold_db >= 0.0f) fl[0].squelch.set_squelch_snr_threshold(cc->squelch_snr_threshold_db);
        /* notch: src/config.cpp:516-564 */
        if (cc->notch_freq > 0) {
            float q = cc->notch_q > 0 ? cc->notch_q : 10.0f;
            fl[0].notch_filter = NotchFilter(cc->notch_freq, WAVE_RATE, q);
        }
        /* ctcss: src/config.cpp:565-591 */
        if (cc->ctcss_freq > 0) fl[0].squelch.set_ctcss_freq(cc->ctcss_freq, WAVE_RATE);
        /* bandwidth: src/config.cpp:592-619 */
        if (cc->bandwidth_hz != 0) {
            channel->needs_raw_iq = 1;
            if (cc->bandwidth_hz > 0) fl[0].lowpass_filter = LowpassFilter((float)cc->bandwidth_hz / 2, WAVE_RATE);
        }
        fl[0].ampfactor = cc->ampfactor; /* src/config.cpp:620-645 */
#ifdef NFM
        if (cc->tau_us >= 0) channel->alpha = (cc->tau_us == 0 ? 0.0f : exp(-1.0f / (WAVE_RATE * 1e-6 * cc->tau_us)));
#endif
#ifdef WITH_AIRBAND_HIP
        /* what the patched parse_channels() keeps through demod_hip_keep_channel_cfg() (integration/demod_hip.cpp): the numbers
         * as they stand in the config file */
        channel->cfg_squelch_threshold = cc->squelch_threshold_dbfs;
        channel->cfg_squelch_snr = cc->squelch_snr_threshold_db;
        channel->cfg_notch = cc->notch_freq;
        channel->cfg_notch_q = cc->notch_q;
        channel->cfg_ctcss = cc->ctcss_freq;
        channel->cfg_bandwidth = cc->bandwidth_hz;
        channel->cfg_tau = cc->tau_us;
#endif
        /* raw-I/Q output: parse_outputs() sets both flags for a rawfile output (src/config.cpp:34-263) */
        if (cc->has_iq_outputs) {
            channel->has_iq_outputs = 1;
            channel->needs_raw_iq = 1;
        }
        /* bins: src/config.cpp:666-667 */
        dev->base_bins[j] = dev->bins[j] =
            (size_t)ceil((fl[0].frequency + input->sample_rate - input->centerfreq) / (double)(input->sample_rate / fft_size) - 1.0) % fft_size;
#ifdef NFM
        if (fl[0].modulation == MOD_NFM) channel->needs_raw_iq = 1; /* src/config.cpp:670-677 */
#endif
        if (channel->needs_raw_iq) { /* src/config.cpp:679-712 */
            double dm_dphi = (double)(fl[0].frequency - input->centerfreq);
            double decimation_factor = ((double)input->sample_rate / (double)WAVE_RATE);
            double dm_dphi_correction = (double)WAVE_RATE / 2.0;
            dm_dphi_correction *= (decimation_factor - round(decimation_factor));
            dm_dphi_correction *= (double)(fl[0].frequency - input->centerfreq) / ((double)input->sample_rate / 2.0);
            dm_dphi -= dm_dphi_correction;
            dm_dphi /= (double)WAVE_RATE;
            dm_dphi -= trunc(dm_dphi);
            dm_dphi *= 256.0 * 65536.0;
            channel->dm_dphi = (uint32_t)((int)dm_dphi);
            channel->dm_phi = 0.f;
        }
#ifdef DEBUG_SQUELCH
        if (g_trace_dir[0]) {
            char path[1024];
            snprintf(path, sizeof(path), "%s/squelch_debug-%d-%d.dat", g_trace_dir, d, j);
            fl[0].squelch.set_debug_file(path);
        }
#endif
    }
    return 0;
}

/* ---- the drop-in, for real (REFH_PATCHED builds only): this file is then compiled against a scratch copy of the reference with
 * integration/airband_hip.patch applied (-DWITH_AIRBAND_HIP), next to integration/demod_hip.cpp compiled VERBATIM -- the translation
 * unit the patch adds to the reference.  demod_hip.cpp calls the airband_hip_* C ABI directly; so that the oracle build never links
 * against the product, the symbols it needs are provided here as trampolines into a dlopen()ed libairband_hip.so. --------------- */
#ifdef REFH_PATCHED
struct HipApi {
    void* dl;
    int (*prepare)(const airband_hip_config*, airband_hip_handle**);
    void (*release)(airband_hip_handle*);
    int (*get_geometry)(const airband_hip_handle*, airband_hip_geometry*);
    const char* (*last_error)(const airband_hip_handle*);
    int64_t (*submit)(airband_hip_handle*, int32_t, const void*, size_t);
    int (*process)(airband_hip_handle*);
    int (*collect)(airband_hip_handle*, float*, float*, char*, airband_hip_channel_stats*);
    int (*device_enable)(airband_hip_handle*, int32_t, int32_t);
    int (*gpu_count)(void);
    int (*batch_ready)(airband_hip_handle*);
    int (*set_mixers)(airband_hip_handle*, int32_t, const airband_hip_mixer_input*, int32_t);
    int (*mixer_set_stereo)(airband_hip_handle*, int32_t, int32_t);
    int (*collect_mixers)(airband_hip_handle*, float*, float*, uint8_t*);
    int (*comm_init_all)(airband_hip_handle**, int32_t);
    int (*comm_group_begin)(void);
    int (*comm_group_end)(void);
    int (*allreduce_mixers)(airband_hip_handle*, void*);
    int (*add_mixers)(airband_hip_handle*, airband_hip_handle*);
    int (*clear_mixers)(airband_hip_handle*);
};
static HipApi g_hip;

extern "C" {
int airband_hip_prepare(const airband_hip_config* cfg, airband_hip_handle** out) { return g_hip.prepare(cfg, out); }
void airband_hip_release(airband_hip_handle* h) { g_hip.release(h); }
int airband_hip_get_geometry(const airband_hip_handle* h, airband_hip_geometry* g) { return g_hip.get_geometry(h, g); }
const char* airband_hip_last_error(const airband_hip_handle* h) { return g_hip.last_error(h); }
int64_t airband_hip_submit(airband_hip_handle* h, int32_t dev, const void* iq, size_t n) { return g_hip.submit(h, dev, iq, n); }
int airband_hip_process(airband_hip_handle* h) { return g_hip.process(h); }
int airband_hip_collect(airband_hip_handle* h, float* w, float* q, char* a, airband_hip_channel_stats* st) { return g_hip.collect(h, w, q, a, st); }
int airband_hip_device_enable(airband_hip_handle* h, int32_t dev, int32_t on) { return g_hip.device_enable(h, dev, on); }
int airband_hip_gpu_count(void) { return g_hip.gpu_count(); }
int airband_hip_batch_ready(airband_hip_handle* h) { return g_hip.batch_ready(h); }
int airband_hip_set_mixers(airband_hip_handle* h, int32_t n, const airband_hip_mixer_input* in, int32_t k) { return g_hip.set_mixers(h, n, in, k); }
int airband_hip_mixer_set_stereo(airband_hip_handle* h, int32_t m, int32_t on) { return g_hip.mixer_set_stereo(h, m, on); }
int airband_hip_collect_mixers(airband_hip_handle* h, float* l, float* r, uint8_t* s) { return g_hip.collect_mixers(h, l, r, s); }
int airband_hip_comm_init_all(airband_hip_handle** hs, int32_t n) { return g_hip.comm_init_all(hs, n); }
int airband_hip_comm_group_begin(void) { return g_hip.comm_group_begin(); }
int airband_hip_comm_group_end(void) { return g_hip.comm_group_end(); }
int airband_hip_allreduce_mixers(airband_hip_handle* h, void* s) { return g_hip.allreduce_mixers(h, s); }
int airband_hip_add_mixers(airband_hip_handle* a, airband_hip_handle* b) { return g_hip.add_mixers(a, b); }
int airband_hip_clear_mixers(airband_hip_handle* h) { return g_hip.clear_mixers(h); }
}

/* statistics exactly as the stats file / TUI would read them with the HIP backend: through the reference's own Squelch getters
 * (src/output.cpp:617-761), which the shim feeds through the patch's Squelch::mirror() */
int refh_hip_channel_stats(int d, int j, airband_hip_channel_stats* out) {
    if (d < 0 || d >= device_count || j < 0 || j >= devices[d].channel_count) return -1;
    freq_t* f = devices[d].channels[j].freqlist;
    memset(out, 0, sizeof(*out));
    out->noise_level = f->squelch.noise_level();
    out->signal_level = f->squelch.signal_level();
    out->squelch_level = f->squelch.squelch_level();
    out->agcavgfast = f->agcavgfast;
    out->open_count = f->squelch.open_count();
    out->flappy_count = f->squelch.flappy_count();
    out->ctcss_count = f->squelch.ctcss_count();
    out->no_ctcss_count = f->squelch.no_ctcss_count();
    out->active_counter = f->active_counter;
    out->bin = (int32_t)devices[d].bins[j];
    out->signal_outside_filter = f->squelch.signal_outside_filter() ? 1 : 0; /* through the patch's mirror */
    return 0;
}

/* Starts ONE demodulate_hip() thread (integration/demod_hip.cpp) over all devices, instead of demodulate(). */
int refh_start_hip(const char* lib_path) {
    g_hip.dl = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
    if (!g_hip.dl) {
        fprintf(stderr, "refh_start_hip: %s\n", dlerror());
        return -1;
    }
#define REFH_SYM(field, name)                                   \
    *(void**)(&g_hip.field) = dlsym(g_hip.dl, "airband_hip_" name); \
    if (!g_hip.field) return -2;
    REFH_SYM(prepare, "prepare") REFH_SYM(release, "release") REFH_SYM(get_geometry, "get_geometry") REFH_SYM(last_error, "last_error")
    REFH_SYM(submit, "submit") REFH_SYM(process, "process") REFH_SYM(collect, "collect") REFH_SYM(device_enable, "device_enable") REFH_SYM(gpu_count, "gpu_count")
    REFH_SYM(batch_ready, "batch_ready") REFH_SYM(set_mixers, "set_mixers") REFH_SYM(mixer_set_stereo, "mixer_set_stereo") REFH_SYM(collect_mixers, "collect_mixers")
    REFH_SYM(comm_init_all, "comm_init_all") REFH_SYM(comm_group_begin, "comm_group_begin") REFH_SYM(comm_group_end, "comm_group_end")
    REFH_SYM(allreduce_mixers, "allreduce_mixers") REFH_SYM(add_mixers, "add_mixers") REFH_SYM(clear_mixers, "clear_mixers")
#undef REFH_SYM
    devices_running = device_count;
    g_demod_params[0].mp3_signal = &g_signal;
    g_demod_params[0].device_start = 0;
    g_demod_params[0].device_end = device_count;
    if (pthread_create(&g_demod_thread[0], NULL, &demodulate_hip, &g_demod_params[0]) != 0) return -1;
    g_threads_running = 1;
    return 0;
}
#else
int refh_hip_channel_stats(int, int, airband_hip_channel_stats*) { return -1; }
int refh_start_hip(const char*) { return -3; } /* only the *_patched builds carry the HIP backend */
#endif /* REFH_PATCHED */

/* ---- mixers: what parse_mixers() / parse_outputs() build (src/config.cpp:835-880, the "mixer" output type :160-190), by hand -------- */
static char g_rx_started[4096]; /* devices whose input driver's own rx thread runs (file inputs) */
static Signal g_mixer_signal; /* the Signal of the output thread that serves the mixers (src/rtl_airband.cpp:1097-1100) */
static pthread_t g_mixer_thread;
static int g_mixer_thread_running = 0;

int refh_add_mixers(int n) {
    mixers = (mixer_t*)XCALLOC(n, sizeof(mixer_t));
    mixer_count = n;
    for (int m = 0; m < n; m++) {
        mixer_t* mixer = mixers + m;
        char name[32];
        snprintf(name, sizeof(name), "mixer%d", m);
        mixer->name = strdup(name);
        mixer->enabled = false;
        mixer->interval = MIX_DIVISOR;
        mixer->channel.highpass = 100;
        mixer->channel.lowpass = 2500;
        mixer->channel.mode = MM_MONO;
        mixer->channel.state = CH_DIRTY;
        mixer->channel.axcindicate = NO_SIGNAL;
    }
    return 0;
}

/* channel (d, j) feeds mixer m: one more output of type O_MIXER on the channel, one more input on the mixer (mixer_connect_input,
 * src/mixer.cpp:57-94).  Returns the mixer's input index. */
int refh_connect(int d, int j, int m, float ampfactor, float balance) {
    if (d < 0 || d >= device_count || j < 0 || j >= devices[d].channel_count || m < 0 || m >= mixer_count) return -1;
    channel_t* channel = devices[d].channels + j;
    const int input = mixer_connect_input(mixers + m, ampfactor, balance);
    if (input < 0) return -2;
    channel->outputs = (output_t*)XREALLOC(channel->outputs, (channel->output_count + 1) * sizeof(output_t));
    output_t* o = channel->outputs + channel->output_count++;
    memset(o, 0, sizeof(*o));
    o->type = O_MIXER;
    o->enabled = true;
    mixer_data* md = (mixer_data*)XCALLOC(1, sizeof(mixer_data));
    md->mixer = mixers + m;
    md->input = input;
    o->data = md;
    return input;
}

int refh_start_mixer_thread(void) {
    if (mixer_count <= 0 || g_mixer_thread_running) return 0;
    if (pthread_create(&g_mixer_thread, NULL, &mixer_thread, &g_mixer_signal) != 0) return -1;
    g_mixer_thread_running = 1;
    return 0;
}

/* HIP backend only: waits until demodulate_hip() has marked every enabled mixer gpu_served (it does so before it first calls into the HIP runtime) or the
 * time is up; returns how many are marked.  The harness starts mixer_thread() after this: the reference starts it before the demodulators
 * (src/rtl_airband.cpp:1098-1111) and emits silence until they deliver, which is timing, not behaviour -- a comparison batch by batch must not see it. */
int refh_wait_mixers_served(double timeout_s) {
    int served = 0;
#ifdef REFH_PATCHED
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    for (;;) {
        int enabled = 0;
        served = 0;
        for (int m = 0; m < mixer_count; m++) {
            if (!mixers[m].enabled) continue;
            enabled++;
            if (mixers[m].gpu_served) served++;
        }
        if (served == enabled) break;
        gettimeofday(&t1, NULL);
        if ((t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec) > timeout_s) break;
        usleep(1000);
    }
#else
    (void)timeout_s;
#endif
    return served;
}

void refh_mixer_counters(int m, uint64_t* out4) {
    out4[0] = mixers[m].output_overrun_count;
    out4[1] = mixers[m].enabled ? 1 : 0;
    out4[2] = (uint64_t)mixers[m].input_count;
#ifdef WITH_AIRBAND_HIP
    out4[3] = mixers[m].gpu_served ? 1 : 0;
#else
    out4[3] = 0;
#endif
}

/* ---- the reference's own file input (src/input-file.cpp:82-181) in place of the hand-built input of refh_add_device(): file_input_new(),
 * the two values file_parse_config() reads from the config, the ring as parse_devices() sizes it (src/config.cpp:796-806), then input_init() and
 * input_start() (src/input-common.cpp:56-87) -- the rx thread replays the file paced by speedup_factor and sets INPUT_FAILED at end of file. */
int refh_use_file_input(int d, const char* path, float speedup_factor) {
    if (d < 0 || d >= device_count) return -1;
    device_t* dev = devices + d;
    input_t* old = dev->input;
    input_t* in = file_input_new();
    file_dev_data_t* dd = (file_dev_data_t*)in->dev_data;
    dd->filepath = strdup(path);
    dd->speedup_factor = speedup_factor;
    in->sample_rate = old->sample_rate;
    in->centerfreq = old->centerfreq;
    in->buf_size = old->buf_size;
    in->buffer = old->buffer;
    in->bufs = in->bufe = 0;
    in->overflow_count = 0;
    dev->input = in;
    free(old);
    if (input_init(in) != 0) return -2;
    return 0;
}
int refh_start_inputs(void) {
    for (int d = 0; d < device_count; d++)
        if (devices[d].input->state == INPUT_INITIALIZED) {
            if (input_start(devices[d].input) != 0) return -1;
            if (d < 4096) g_rx_started[d] = 1;
        }
    return 0;
}

/* ---- the waterfall (src/rtl_airband.cpp:632-643): `tui` on, stdout into a file for the duration ------------------------------------ */
static int g_saved_stdout = -1;
int refh_tui_begin(const char* path) {
    fflush(stdout);
    g_saved_stdout = dup(1);
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    dup2(fileno(f), 1);
    fclose(f);
    tui = 1;
    return 0;
}
void refh_tui_end(void) {
    tui = 0;
    fflush(stdout);
    if (g_saved_stdout >= 0) {
        dup2(g_saved_stdout, 1);
        close(g_saved_stdout);
        g_saved_stdout = -1;
    }
}

/* n_threads demodulate() instances over contiguous device shards: the reference's own
 * multiple_demod_threads model (src/rtl_airband.cpp:1052-1086,1110-1112), 1 = its default. */
int refh_start(int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > device_count) n_threads = device_count;
    if (n_threads > REFH_MAX_THREADS) n_threads = REFH_MAX_THREADS;
    devices_running = device_count;
    for (int t = 0; t < n_threads; t++) {
        int start = (int)((long)device_count * t / n_threads), end = (int)((long)device_count * (t + 1) / n_threads);
        init_demod(&g_demod_params[t], &g_signal, start, end); /* src/rtl_airband.cpp:253-266 */
        if (pthread_create(&g_demod_thread[t], NULL, &demodulate, &g_demod_params[t]) != 0) return -1;
        g_threads_running = t + 1;
    }
    return 0;
}

static void refh_join_inputs(void) {
    for (int d = 0; d < device_count && d < 4096; d++)
        if (g_rx_started[d]) {
            pthread_join(devices[d].input->rx_thread, NULL);
            g_rx_started[d] = 0;
        }
}
void refh_stop(void) {
    do_exit = 1;
    refh_join_inputs();
    if (!g_threads_running && !g_mixer_thread_running) return;
    for (int t = 0; t < g_threads_running; t++) pthread_join(g_demod_thread[t], NULL);
    g_threads_running = 0;
    if (g_mixer_thread_running) {
        pthread_join(g_mixer_thread, NULL);
        g_mixer_thread_running = 0;
    }
#ifdef DEBUG_SQUELCH
    for (int d = 0; d < device_count; d++)
        for (int j = 0; j < devices[d].channel_count; j++) devices[d].channels[j].freqlist[0].squelch.~Squelch(); /* flush traces */
#endif
}

static size_t refh_available(input_t* in) {
    pthread_mutex_lock(&in->buffer_lock);
    size_t a = in->bufe >= in->bufs ? in->bufe - in->bufs : in->buf_size - in->bufs + in->bufe;
    pthread_mutex_unlock(&in->buffer_lock);
    return a;
}

static size_t refh_bps(input_t* in) { return 2 * in->bytes_per_sample * (size_t)round((double)in->sample_rate / (double)WAVE_RATE); }

/* Role of output_thread (src/output.cpp:903-923) for one device: copy results out, tail copy, clear flag. */
static void refh_drain(int d, float* waveout, float* iq_out, char* axc) {
    device_t* dev = devices + d;
    for (int j = 0; j < dev->channel_count; j++) {
        channel_t* channel = dev->channels + j;
        if (waveout) memcpy(waveout + (size_t)j * WAVE_BATCH, channel->waveout, sizeof(float) * WAVE_BATCH);
        if (iq_out) memcpy(iq_out + (size_t)j * 2 * WAVE_BATCH, channel->iq_out, sizeof(float) * 2 * WAVE_BATCH);
        if (axc) axc[j] = (char)channel->axcindicate;
        for (int k = 0; k < channel->output_count; k++) /* process_outputs(), src/output.cpp:533-535 */
            if (channel->outputs[k].type == O_MIXER && channel->outputs[k].enabled) {
                mixer_data* mdata = (mixer_data*)(channel->outputs[k].data);
                mixer_put_samples(mdata->mixer, mdata->input, channel->waveout, channel->axcindicate != NO_SIGNAL, WAVE_BATCH);
            }
        memcpy(channel->waveout, channel->waveout + WAVE_BATCH, AGC_EXTRA * 4); /* src/output.cpp:920 */
    }
    dev->waveavail = 0;
}

/* Streams `nbytes` of I/Q into device d and collects every batch that completes.
 * waveout [max_batches][channel_count][WAVE_BATCH], iq_out [max_batches][channel_count][2*WAVE_BATCH],
 * axc [max_batches][channel_count].  Returns the number of batches produced. */
int refh_run_device(int d, const unsigned char* iq, size_t nbytes, int max_batches, float* waveout, float* iq_out, char* axc) {
    device_t* dev = devices + d;
    input_t* in = dev->input;
    const size_t bps = refh_bps(in);
    const size_t need = bps * FFT_BATCH + fft_size * in->bytes_per_sample * 2;
    const int C = dev->channel_count;
    int nb = 0;
    size_t off = 0;
    while (off < nbytes || refh_available(in) >= need || dev->waveavail) {
        if (dev->waveavail) {
            if (nb >= max_batches) break;
            refh_drain(d, waveout ? waveout + (size_t)nb * C * WAVE_BATCH : NULL, iq_out ? iq_out + (size_t)nb * C * 2 * WAVE_BATCH : NULL,
                       axc ? axc + (size_t)nb * C : NULL);
            nb++;
            continue;
        }
        if (refh_available(in) >= need) {
            sched_yield();
            continue;
        }
        /* feed at most one hop-batch at a time so that at most one output batch can complete */
        size_t room = in->buf_size - 1 - refh_available(in);
        size_t n = std::min(nbytes - off, std::min(room, bps * (size_t)WAVE_BATCH / 4));
        if (n == 0) break;
        circbuffer_append(in, const_cast<unsigned char*>(iq) + off, n);
        off += n;
    }
    g_hops_fed_bytes[d] += off;
    return nb;
}

/* Streams I/Q into EVERY device concurrently (round-robin, like independent rx threads) and collects batches until each device produced
 * its target.  Works with demodulate() and with demodulate_hip().  nbytes[d] = bytes of device d's stream; fail_after[d] >= 0 plays
 * the file input at end of file (src/input-file.cpp:101-111): once device d has delivered that many batches its rx side stops and
 * sets input->state = INPUT_FAILED; its target is then fail_after[d] batches instead of n_batches.  fail_after may be NULL.
 * waveout [device][n_batches][C][WAVE_BATCH] with C = the (common) channel count, etc.; got[d] = batches device d produced.
 * Returns the minimum over the devices of (batches produced - target), i.e. 0 when every device reached its target. */
/* where refh_run_all() puts what the mixers emit (the output thread's role for mixer channels, src/output.cpp:887-896): left / right
 * [mixer][n_batches][WAVE_BATCH], axc [mixer][n_batches], got [mixer]; NULL = mixers are not collected */
static float* g_mix_left = NULL;
static float* g_mix_right = NULL;
static char* g_mix_axc = NULL;
static int* g_mix_got = NULL;
void refh_set_mixer_outputs(float* left, float* right, char* axc, int* got) {
    g_mix_left = left;
    g_mix_right = right;
    g_mix_axc = axc;
    g_mix_got = got;
}

int refh_run_all(const unsigned char* const* iq, const size_t* nbytes, const int* fail_after, int n_batches, float* waveout, float* iq_out, char* axc, int* got,
                 double timeout_s) {
    std::vector<size_t> off(device_count, 0);
    std::vector<int> nb(device_count, 0), target(device_count, n_batches);
    for (int d = 0; d < device_count; d++)
        if (fail_after && fail_after[d] >= 0 && fail_after[d] < n_batches) target[d] = fail_after[d];
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    const int C = devices[0].channel_count;
    std::vector<int> mb(mixer_count, 0);
    for (;;) {
        bool done = true;
        /* mixers first: a mixer channel that is CH_READY is taken like output_thread() takes it; devices are not fed beyond the batch the slowest
         * enabled mixer still has to emit (mixer_thread() looks at its inputs every 1/16 s: a device that ran ahead would overrun them) */
        int mixers_at = n_batches;
        for (int m = 0; m < mixer_count; m++) {
            mixer_t* mixer = mixers + m;
            if (!mixer->enabled) continue;
            channel_t* channel = &mixer->channel;
            if (channel->state == CH_READY) {
                if (g_mix_left && mb[m] < n_batches) {
                    const size_t o = (size_t)m * n_batches + mb[m];
                    memcpy(g_mix_left + o * WAVE_BATCH, channel->waveout, sizeof(float) * WAVE_BATCH);
                    if (g_mix_right) {
                        if (channel->mode == MM_STEREO) memcpy(g_mix_right + o * WAVE_BATCH, channel->waveout_r, sizeof(float) * WAVE_BATCH);
                        else memset(g_mix_right + o * WAVE_BATCH, 0, sizeof(float) * WAVE_BATCH);
                    }
                    if (g_mix_axc) g_mix_axc[o] = (char)channel->axcindicate;
                }
                mb[m]++;
                channel->state = CH_DIRTY;
            }
            if (g_mix_left) {
                mixers_at = std::min(mixers_at, mb[m]);
                if (mb[m] < n_batches) {
                    bool feeders_left = false; /* a mixer all of whose devices have reached their target emits nothing more */
                    for (int d = 0; d < device_count; d++)
                        if (nb[d] < target[d] || nb[d] > mb[m]) feeders_left = true;
                    if (feeders_left) done = false;
                }
            }
        }
        for (int d = 0; d < device_count; d++) {
            device_t* dev = devices + d;
            input_t* in = dev->input;
            if (g_mix_left && mixer_count > 0 && nb[d] > mixers_at) { /* wait for the mixers to emit this device's last batch */
                if (nb[d] < target[d]) done = false;
                continue;
            }
            if (dev->waveavail && nb[d] < target[d]) {
                const size_t o = ((size_t)d * n_batches + nb[d]) * C;
                refh_drain(d, waveout ? waveout + o * WAVE_BATCH : NULL, iq_out ? iq_out + o * 2 * WAVE_BATCH : NULL, axc ? axc + o : NULL);
                nb[d]++;
                if (nb[d] == target[d] && target[d] < n_batches && in->state == INPUT_RUNNING) in->state = INPUT_FAILED; /* "hit end of file ... disabling" */
            }
            /* Never queue more than the batch that is about to be drained needs: the reference's hand-off (waveavail flag, tail
             * copy by the consumer, src/output.cpp:917-922) is racy by design when the demod thread can run a whole batch ahead of
             * the output thread, and a deterministic harness must not depend on who wins. */
            const size_t cap = ((size_t)(nb[d] + 1) * WAVE_BATCH + AGC_EXTRA + 1) * refh_bps(in) + fft_size * in->bytes_per_sample * 2;
            if (nb[d] < target[d] && off[d] < nbytes[d] && off[d] < cap) {
                const size_t room = in->buf_size - 1 - refh_available(in);
                const size_t n = std::min(std::min(nbytes[d], cap) - off[d], std::min(room, refh_bps(in) * (size_t)WAVE_BATCH / 4));
                if (n > 0) {
                    circbuffer_append(in, const_cast<unsigned char*>(iq[d]) + off[d], n);
                    off[d] += n;
                }
            }
            /* an input driver of the reference's own that has ended (file input at end of file -> INPUT_FAILED -> demodulate() disables the device):
             * what it delivered is all there will be */
            if (nb[d] < target[d] && g_rx_started[d < 4096 ? d : 0] && d < 4096 && in->state != INPUT_RUNNING && in->state != INPUT_INITIALIZED && !dev->waveavail &&
                refh_available(in) < refh_bps(in) * (size_t)WAVE_BATCH)
                target[d] = nb[d];
            if (nb[d] < target[d]) done = false;
        }
        if (done) break;
        gettimeofday(&t1, NULL);
        if ((t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec) > timeout_s) break;
        sched_yield();
    }
    int mn = 0;
    for (int d = 0; d < device_count; d++) {
        if (got) got[d] = nb[d];
        mn = std::min(mn, nb[d] - target[d]);
    }
    if (g_mix_got)
        for (int m = 0; m < mixer_count; m++) g_mix_got[m] = mb[m];
    return mn;
}

/* End of every stream: all inputs report INPUT_FAILED (file inputs at end of file).  The demodulator -- demodulate() or demodulate_hip() --
 * must take each device out (disable_device_outputs, devices_running--) and, with none left, set do_exit and return
 * (src/rtl_airband.cpp:377-391).  Returns 1 if the thread(s) ended on their own within timeout_s, 0 otherwise (they are then stopped). */
int refh_fail_all_and_wait_exit(double timeout_s) {
    for (int d = 0; d < device_count; d++)
        if (devices[d].input->state == INPUT_RUNNING) devices[d].input->state = INPUT_FAILED;
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    while (!do_exit) {
        gettimeofday(&t1, NULL);
        if ((t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec) > timeout_s) return 0;
        sched_yield();
    }
    for (int t = 0; t < g_threads_running; t++) pthread_join(g_demod_thread[t], NULL);
    g_threads_running = 0;
    if (g_mixer_thread_running) {
        pthread_join(g_mixer_thread, NULL);
        g_mixer_thread_running = 0;
    }
    return 1;
}

/* inputs that end on their own (file inputs at end of file): waits for the demodulator to notice that every receiver has failed */
int refh_wait_exit(double timeout_s) {
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    while (!do_exit) {
        gettimeofday(&t1, NULL);
        if ((t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec) > timeout_s) return 0;
        usleep(1000);
    }
    for (int t = 0; t < g_threads_running; t++) pthread_join(g_demod_thread[t], NULL);
    g_threads_running = 0;
    if (g_mixer_thread_running) {
        pthread_join(g_mixer_thread, NULL);
        g_mixer_thread_running = 0;
    }
    return 1;
}

int refh_outputs_disabled(int d) { return (d >= 0 && d < 4096) ? g_outputs_disabled[d] : -1; }
int refh_devices_running(void) { return devices_running; }
int refh_input_state(int d) { return (int)devices[d].input->state; }
unsigned refh_output_overruns(int d) { return (unsigned)devices[d].output_overrun_count; }

/* stats mirror: what src/output.cpp:617-761 and the TUI read */
int refh_channel_stats(int d, int j, airband_hip_channel_stats* out) {
    device_t* dev = devices + d;
    channel_t* channel = dev->channels + j;
    freq_t* f = channel->freqlist;
    out->noise_level = f->squelch.noise_level();
    out->signal_level = f->squelch.signal_level();
    out->squelch_level = f->squelch.squelch_level();
    out->agcavgfast = f->agcavgfast;
    out->open_count = f->squelch.open_count();
    out->flappy_count = f->squelch.flappy_count();
    out->ctcss_count = f->squelch.ctcss_count();
    out->no_ctcss_count = f->squelch.no_ctcss_count();
    out->active_counter = f->active_counter;
    out->bin = (int32_t)dev->bins[j];
    out->squelch_state = -1; /* private in the reference */
    out->signal_outside_filter = f->squelch.signal_outside_filter() ? 1 : 0;
    out->reserved = 0;
    return 0;
}

/* derived constants, same slot meaning as airband_hip_channel_constants() where the reference exposes them */
int refh_channel_constants(int d, int j, double* v) {
    device_t* dev = devices + d;
    channel_t* channel = dev->channels + j;
    for (int i = 0; i < 16; i++) v[i] = NAN;
    v[0] = (double)dev->bins[j];
    v[1] = (double)channel->dm_dphi;
#ifdef NFM
    v[2] = (double)channel->alpha;
#endif
    v[13] = (double)channel->needs_raw_iq;
    return 0;
}

/* ---- stand-alone pieces of the reference, exported for unit-level pinning of the restatement ---- */

/* Squelch driven exactly like src/test_squelch.cpp drives it: raw samples only. */
void* refh_squelch_new(float snr_db, int manual_dbfs, float ctcss_freq) {
    Squelch* s = new Squelch();
    if (manual_dbfs < 0) s->set_squelch_level_threshold(dBFS_to_level((float)manual_dbfs));
    if (snr_db >= 0) s->set_squelch_snr_threshold(snr_db);
    if (ctcss_freq > 0) s->set_ctcss_freq(ctcss_freq, WAVE_RATE);
    return s;
}
void refh_squelch_raw(void* p, const float* x, int n, unsigned char* is_open, float* noise, float* level) {
    Squelch* s = (Squelch*)p;
    for (int i = 0; i < n; i++) {
        s->process_raw_sample(x[i]);
        if (is_open) is_open[i] = (unsigned char)((s->is_open() ? 1 : 0) | (s->should_process_audio() ? 2 : 0) | (s->should_filter_sample() ? 4 : 0) |
                                                  (s->first_open_sample() ? 8 : 0) | (s->last_open_sample() ? 16 : 0));
        if (noise) noise[i] = s->noise_level();
        if (level) level[i] = s->squelch_level();
    }
}
/* raw + filtered path in the order demodulate() calls them (src/rtl_airband.cpp:507,510,526) */
void refh_squelch_raw_filtered(void* p, const float* raw, const float* filtered, int n, unsigned char* is_open, float* noise, float* level) {
    Squelch* s = (Squelch*)p;
    for (int i = 0; i < n; i++) {
        s->process_raw_sample(raw[i]);
        if (s->should_filter_sample()) s->process_filtered_sample(filtered[i]);
        if (is_open) is_open[i] = (unsigned char)((s->is_open() ? 1 : 0) | (s->should_process_audio() ? 2 : 0) | (s->should_filter_sample() ? 4 : 0) |
                                                  (s->first_open_sample() ? 8 : 0) | (s->last_open_sample() ? 16 : 0));
        if (noise) noise[i] = s->noise_level();
        if (level) level[i] = s->squelch_level();
    }
}
/* raw + audio path as the ctcss squelch tests do (src/test_squelch.cpp:167-281) */
void refh_squelch_raw_audio(void* p, const float* raw, const float* audio, int n, unsigned char* is_open) {
    Squelch* s = (Squelch*)p;
    for (int i = 0; i < n; i++) {
        s->process_raw_sample(raw[i]);
        if (s->should_process_audio()) s->process_audio_sample(audio[i]);
        if (is_open) is_open[i] = (unsigned char)((s->is_open() ? 1 : 0) | (s->should_process_audio() ? 2 : 0));
    }
}
/* audio first, then raw, unconditionally: the call order of the reference's own CTCSS squelch tests
 * (src/test_squelch.cpp:186-189) */
void refh_squelch_audio_raw(void* p, const float* raw, const float* audio, int n, unsigned char* is_open) {
    Squelch* s = (Squelch*)p;
    for (int i = 0; i < n; i++) {
        s->process_audio_sample(audio[i]);
        s->process_raw_sample(raw[i]);
        if (is_open) is_open[i] = (unsigned char)((s->is_open() ? 1 : 0) | (s->should_process_audio() ? 2 : 0));
    }
}
void refh_squelch_counts(void* p, uint64_t* out4) {
    Squelch* s = (Squelch*)p;
    out4[0] = s->open_count();
    out4[1] = s->flappy_count();
    out4[2] = s->ctcss_count();
    out4[3] = s->no_ctcss_count();
}
void refh_squelch_free(void* p) { delete (Squelch*)p; }

/* CTCSS detector alone (src/test_ctcss.cpp) */
void refh_ctcss_run(float ctcss_freq, float sample_rate, int window, const float* x, int n, unsigned char* has_tone, uint64_t* counts2) {
    CTCSS c(ctcss_freq, sample_rate, window);
    for (int i = 0; i < n; i++) {
        c.process_audio_sample(x[i]);
        if (has_tone) has_tone[i] = (unsigned char)((c.has_tone() ? 1 : 0) | (c.enough_samples() ? 2 : 0));
    }
    counts2[0] = c.found_count();
    counts2[1] = c.not_found_count();
}

float refh_tone_coeff(float f, float rate, int window) {
    ToneDetector t(f, rate, window);
    return t.coefficient();
}

/* filters alone */
void refh_notch_run(float freq, float rate, float q, float* x, int n) {
    NotchFilter f(freq, rate, q);
    for (int i = 0; i < n; i++) f.apply(x[i]);
}
void refh_lowpass_run(float freq, float rate, float* re, float* im, int n) {
    LowpassFilter f(freq, rate);
    for (int i = 0; i < n; i++) f.apply(re[i], im[i]);
}
void refh_sincos_lut(uint32_t phi, float* s, float* c) { sincosf_lut(phi, s, c); }
float refh_dbfs_to_level(float dbfs) { return dBFS_to_level(dbfs); }
#ifdef NFM
float refh_fast_atan2(float y, float x) { return fast_atan2(y, x); }
float refh_polar_disc_fast(float ar, float aj, float br, float bj) { return polar_disc_fast(ar, aj, br, bj); }
float refh_fm_quadri_demod(float ar, float aj, float br, float bj) { return fm_quadri_demod(ar, aj, br, bj); }
#endif

/* ---- throughput mode (CPU baseline) ------------------------------------------------
 * Ring of device d is pre-filled once with `iq` (buf_size bytes + tail) and kept "full" by moving the write cursor, so the timed region
 * contains no memcpy: only demodulate() and the consumers.  The consumer side imitates output_thread (src/output.cpp:903-923) with ONE
 * DRAIN THREAD PER `devices_per_drain` DEVICES (round 4; before: one thread walked every device's buffer_lock in a sched_yield() spin and
 * became the bottleneck at 256 demodulate() threads): it takes a batch when waveavail is up, refills the ring cursor and sleeps 250 us
 * between rounds.  demodulate() never waits for its consumer -- a batch finished while the previous one is still flagged is counted in
 * output_overrun_count (src/rtl_airband.cpp:649-654) and is work done all the same: out[0] = batches drained, out[1] = batches that
 * overran, both over all devices; their sum is what the demodulate() threads completed in out[2] seconds of wall time. */
struct tp_drain_t {
    int d0, d1;
    volatile int stop;
    long drained;
    pthread_t thread;
};
static void* tp_drain_main(void* p) {
    tp_drain_t* a = (tp_drain_t*)p;
    while (!a->stop) {
        for (int d = a->d0; d < a->d1; d++) {
            input_t* in = devices[d].input;
            if (devices[d].waveavail) {
                refh_drain(d, NULL, NULL, NULL);
                a->drained++;
            }
            /* keep the ring full: write cursor trails the read cursor by one hop */
            pthread_mutex_lock(&in->buffer_lock);
            in->bufe = (in->bufs + in->buf_size - refh_bps(in)) % in->buf_size;
            pthread_mutex_unlock(&in->buffer_lock);
        }
        usleep(250);
    }
    return NULL;
}
long refh_throughput(const unsigned char* const* iq_per_device, double seconds, int devices_per_drain, double* out3) {
    if (devices_per_drain < 1) devices_per_drain = 1;
    for (int d = 0; d < device_count; d++) {
        input_t* in = devices[d].input;
        memcpy(in->buffer, iq_per_device[d], in->buf_size + 2 * in->bytes_per_sample * fft_size);
        pthread_mutex_lock(&in->buffer_lock);
        in->bufs = 0;
        in->bufe = in->buf_size - refh_bps(in);
        pthread_mutex_unlock(&in->buffer_lock);
    }
    const int n_drain = (device_count + devices_per_drain - 1) / devices_per_drain;
    std::vector<tp_drain_t> drains(n_drain);
    struct timeval t0, t1;
    /* the demodulate() threads have been running since refh_start() (spinning on empty rings in 10 ms sleeps): what they finished before t0 is taken off */
    std::vector<long> over0(device_count);
    for (int d = 0; d < device_count; d++) over0[d] = (long)devices[d].output_overrun_count;
    gettimeofday(&t0, NULL);
    for (int k = 0; k < n_drain; k++) {
        drains[k].d0 = k * devices_per_drain;
        drains[k].d1 = std::min(device_count, (k + 1) * devices_per_drain);
        drains[k].stop = 0;
        drains[k].drained = 0;
        pthread_create(&drains[k].thread, NULL, tp_drain_main, &drains[k]);
    }
    struct timespec nap;
    nap.tv_sec = (time_t)seconds;
    nap.tv_nsec = (long)((seconds - (double)nap.tv_sec) * 1e9);
    nanosleep(&nap, NULL);
    long overruns = 0;
    for (int d = 0; d < device_count; d++) overruns += (long)devices[d].output_overrun_count - over0[d];
    gettimeofday(&t1, NULL);
    long drained = 0;
    for (int k = 0; k < n_drain; k++) drains[k].stop = 1;
    for (int k = 0; k < n_drain; k++) {
        pthread_join(drains[k].thread, NULL);
        drained += drains[k].drained;
    }
    const double el = (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec);
    if (out3) {
        out3[0] = (double)drained;
        out3[1] = (double)overruns;
        out3[2] = el;
    }
    return drained + overruns;
}

} /* extern "C" */
