// aln_hints.hip — per-context tuning / kernel-selection hints (aln_ctx_set_hint, aln_ctx_get_hint).
//
// Every switch that used to be a getenv() at launch time is a field of aln_hints.  A context takes its defaults from the
// environment ONCE, when it is created (so tools/ab_variants.sh-style A/B runs keep working), and a caller changes them
// explicitly through the C ABI afterwards; no launch reads the environment.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "aln_internal.h"

namespace {

struct HintDef {
  const char* key;        // aln_ctx_set_hint key
  const char* env;        // environment variable read at aln_ctx_create (nullptr: none)
  bool env_negates;       // the variable's presence means 0 (the historical ALN_NO_* switches)
  int aln_hints::*field;
};

const HintDef kDefs[] = {
    {"tag_kernel", "ALN_NO_TAG_KERNEL", true, &aln_hints::tag_kernel},
    {"h16", "ALN_NO_H16", true, &aln_hints::h16},
    {"key16", "ALN_NO_KEY16", true, &aln_hints::key16},
    {"tag_alt_prio", "ALN_TAG_ALT_PRIO", false, &aln_hints::tag_alt_prio},
    {"tag_lag", "ALN_TAG_LAG", false, &aln_hints::tag_lag},
    {"tag_segments", "ALN_TAG_SEGMENTS", false, &aln_hints::tag_segments},
    {"tag_solo", "ALN_TAG_SOLO", false, &aln_hints::tag_solo},
    {"tag_bits", "ALN_TAG_BITS", false, &aln_hints::tag_bits},
    {"tag_occupancy", "ALN_TAG_OCCUPANCY", false, &aln_hints::tag_occupancy},
    {"dp_variant_nw", nullptr, false, &aln_hints::dp_nw},
    {"dp_variant_r", nullptr, false, &aln_hints::dp_r},
    {"dp_variant_x", nullptr, false, &aln_hints::dp_x},
    {"exact_tiles", "ALN_EXACT_NO_TILES", true, &aln_hints::exact_tiles},
    {"exact_literal", "ALN_EXACT_LITERAL", false, &aln_hints::exact_literal},
    {"exact_alt_prio", "ALN_EXACT_ALT_PRIO", false, &aln_hints::exact_alt_prio},
    {"exact_prune", "ALN_EXACT_PRUNE", false, &aln_hints::exact_prune},
    {"exact_wavefront", "ALN_EXACT_WAVEFRONT", false, &aln_hints::exact_wavefront},
    {"exact_debug", "ALN_EXACT_DEBUG", false, &aln_hints::exact_debug},
    {"score_packed", "ALN_SCORE_NO_PACKED", true, &aln_hints::score_packed},
    {"enum_heavy_first", "ALN_ENUM_HEAVY_FIRST", false, &aln_hints::enum_heavy_first},
    {"enum_pool_retries", "ALN_ENUM_POOL_RETRIES", false, &aln_hints::enum_pool_retries},
    {"enum_waves", "ALN_ENUM_WAVES", false, &aln_hints::enum_waves},
    {"enum_debug", "ALN_ENUM_DEBUG", false, &aln_hints::enum_debug},
    {"enum_keep_pools", "ALN_ENUM_KEEP_POOLS", false, &aln_hints::enum_keep_pools},
    {"plane_row_align", "ALN_PLANE_ROW_ALIGN", false, &aln_hints::plane_row_align},
};

}  // namespace

namespace aln {

void hints_from_env(aln_hints* h) {
  for (const HintDef& d : kDefs) {
    if (!d.env) continue;
    const char* e = getenv(d.env);
    if (!e) continue;
    h->*(d.field) = d.env_negates ? 0 : atoi(e);
  }
  if (const char* e = getenv("ALN_DP_VARIANT")) { h->dp_x = 0; sscanf(e, "%d,%d,%d", &h->dp_nw, &h->dp_r, &h->dp_x); }
  if (const char* e = getenv("ALN_ENUM_NODE_CAP")) h->enum_node_cap = strtoll(e, nullptr, 10);
}

}  // namespace aln

extern "C" int aln_ctx_set_hint(aln_ctx* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return ALN_E_ARG;
  if (strcmp(key, "enum_node_cap") == 0) { ctx->hints.enum_node_cap = value; return ALN_OK; }
  for (const HintDef& d : kDefs)
    if (strcmp(key, d.key) == 0) { ctx->hints.*(d.field) = (int)value; return ALN_OK; }
  return ALN_E_ARG;
}

extern "C" int aln_ctx_get_hint(const aln_ctx* ctx, const char* key, int64_t* value) {
  if (!ctx || !key || !value) return ALN_E_ARG;
  if (strcmp(key, "enum_node_cap") == 0) { *value = ctx->hints.enum_node_cap; return ALN_OK; }
  for (const HintDef& d : kDefs)
    if (strcmp(key, d.key) == 0) { *value = ctx->hints.*(d.field); return ALN_OK; }
  return ALN_E_ARG;
}
