/* csrc/params.h -- host-side derivation of every constant the kernels need from an airband_hip_config.
 * Compiled with g++ (not hipcc) so that libm / complex-division behaviour is the platform's, the same the
 * reference's own config-time code would see. */
#ifndef AIRBAND_CSRC_PARAMS_H
#define AIRBAND_CSRC_PARAMS_H

#include <string>
#include <vector>

#include "../../include/airband_hip.h"
#include "common.h"

namespace airband {

struct ToneTable {               /* one CTCSS-enabled channel: two Goertzel banks */
    float coeff[2][AB_MAX_TONES]; /* [0] fast (0.05 s window), [1] slow (0.4 s window) */
    int n[2];
    int window[2];
};

struct Plan {
    int fft_log = 0, fft_size = 0, wave_rate = 0, wave_batch = 0, fm_demod = 0;
    int n_dev = 0, total_ch = 0, max_ch = 0;
    std::vector<DevConst> dev;
    std::vector<int> chan_base;        /* external index of channel 0 of each dongle */
    std::vector<ChanConst> cc;         /* by external channel index */
    std::vector<ChanState> cs0;        /* initial state, by external channel index */
    std::vector<ToneTable> tones;      /* by ct_slot */
    std::vector<float> window;         /* fft_size coefficients (src/rtl_airband.cpp:335-351) */
    std::vector<float> sin_lut, cos_lut; /* 257 entries each (src/util.cpp:105-110) */
    std::vector<float> twiddle;          /* fft_size (cos, sin) pairs of exp(-2 pi i k / fft_size): the wavefront-FFT channelizer's table */
    float lev_u8[256], lev_s8[256];    /* src/rtl_airband.cpp:316-324 */
    /* matrix-core channelizer tables (channelizer_dft.hip); empty unless the configuration qualifies */
    std::vector<int> item_dev, item_group, item_bset; /* channelizer work items: (dongle, group of 8 channels, coefficient-table index) */
    std::vector<int> item_home;                       /* the SHARED table of the item's base bins: what a group with an AFC channel reads while none of its channels has moved */
    std::vector<int8_t> bfrag;         /* [n_bsets][3][fft_size / 32][64][16] */
    std::vector<double> bcorr;         /* [n_bsets][16] */
    std::vector<float> ftab;           /* CF32 dongles (channelizer_f32.hip): [n_shared_bsets][NW pieces][2 fft_size / 4 / NW MFMAs][64 lanes], NW = 4 up to fft_size 512, 8 beyond (kernels.h f32_nw) window x twiddle */
    double b_unscale = 0.0;
    bool b_edge_hi_zero = false;       /* digit 2 is zero in the outer k-steps (fft_size / 256 at either end) of every table */
    int n_bsets = 0, n_shared_bsets = 0; /* coefficient tables in all / those shared between work items (the rest belong to groups with an AFC channel) */
    int n_host_bsets = 0;                /* shared tables [0, n_host_bsets) are in bfrag / bcorr; [n_host_bsets, n_shared_bsets) are built on the device at prepare() time (4 096, AIRBAND_HIP_HOST_TABLES_MAX) */
    std::vector<int> bset_bins;          /* [n_bsets][8] the bin every column pair of a table is built for (-1: unused) */
    int64_t hop_bytes_max = 0;
    bool uniform_hop = true;           /* every dongle has the same sfmt / hop (needed by the batched launch) */
    std::string error;
};

/* Returns 0 or a negative AIRBAND_HIP_E* code (plan.error holds the text). No GPU needed. */
int build_plan(const airband_hip_config* cfg, Plan& plan);

/* Builds the int8 coefficient tables for the matrix-core channelizer.  host_private = false: tables of groups with an AFC channel are left
 * to the device (retune kernel); their slots exist (item_bset, n_bsets) but plan.bfrag / bcorr hold the shared tables only. */
void build_dft_tables(Plan& plan, bool host_private = true);

/* CF32 dongles on the float32 matrix pipe (channelizer_f32.hip): one float table per shared bin set of build_dft_tables() (call that first), window x twiddle
 * evaluated in double and rounded once, in the order the kernel contracts: entry [bset][piece][s][lane] is coefficient (k, column lane & 15) with
 * k = piece * (fft_size / 2) + 16 (s / 4) + 4 (lane >> 4) + s % 4, value k = 2 n + {0: I, 1: Q} of the window */
void build_f32_tables(Plan& plan);

/* exact_math.h: RN(1 / g) if x * r corrected once equals the IEEE x / g for every x (all 2^23 significands are tried, ~70 ms), else 0 */
float div_const_reciprocal(float g);

/* The 16 "derived constants" slots documented at airband_hip_channel_constants(). */
void channel_constants(const Plan& plan, int ext_index, double* out16);
/* host-only: largest error (relative to the RMS of the exact values) of the matrix-core coefficient tables on `windows` pseudo-random windows per work item */
double dft_table_selftest(const Plan& plan, int windows);
double f32_table_selftest(const Plan& plan, int windows);

}  // namespace airband
#endif
