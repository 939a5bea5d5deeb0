/* oracle/shim/config.h -- TEST INFRASTRUCTURE ONLY.
 * Stand-in for the cmake-generated config.h of the reference (template: src/config.h.in).
 * NFM is supplied by the oracle Makefile (-DNFM) so both the AM-only and the NFM build
 * of the reference can be produced from one shim. */
#ifndef _CONFIG_H
#define _CONFIG_H
#define SINCOSF sincosf
#define SHOUT_SET_METADATA shout_set_metadata
#endif
