/* oracle/shim: opaque libshout handle so the reference's data-model header parses. No network. */
#ifndef ORACLE_SHIM_SHOUT_H
#define ORACLE_SHIM_SHOUT_H
struct oracle_shim_shout_ctx;
typedef struct oracle_shim_shout_ctx shout_t;
static inline void shout_init(void) {}
static inline void shout_shutdown(void) {}
#endif
