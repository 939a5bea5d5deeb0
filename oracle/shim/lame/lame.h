/* oracle/shim: opaque LAME handle so the reference's data-model header parses. No encoder. */
#ifndef ORACLE_SHIM_LAME_H
#define ORACLE_SHIM_LAME_H
struct oracle_shim_lame_ctx;
typedef struct oracle_shim_lame_ctx* lame_t;
static inline int lame_close(lame_t) { return 0; }
#endif
