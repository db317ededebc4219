// todo_stubs.cc -- entry points declared in include/aasr.h whose
// implementation has not landed yet.  They fail loudly.
#include "common.h"
using namespace aasr;
#define TODO(name) return fail(AASR_ERR_UNSUPPORTED, name ": not built yet")
extern "C" {
aasr_status aasr_feat_create(const char *, aasr_feat **) { TODO("aasr_feat_create"); }
void aasr_feat_destroy(aasr_feat *) {}
int aasr_feat_dim(const aasr_feat *) { return -1; }
float aasr_feat_frame_rate(const aasr_feat *) { return 0; }
int aasr_feat_sample_rate(const aasr_feat *) { return -1; }
int aasr_feat_module_dim(const aasr_feat *, const char *) { return -1; }
void aasr_feat_halo(const aasr_feat *, int *l, int *r) { if (l) *l = 0; if (r) *r = 0; }
int aasr_feat_last_frame(const aasr_feat *, int64_t) { return -1; }
aasr_status aasr_feat_run(aasr_feat *, const int16_t *, int64_t, int32_t, int32_t, const char *, float *) { TODO("aasr_feat_run"); }
aasr_status aasr_feat_run_dev(aasr_feat *, const int16_t *, int64_t, int32_t, int32_t, float *, void *) { TODO("aasr_feat_run_dev"); }
aasr_status aasr_feat_run_f64(aasr_feat *, const int16_t *, int64_t, int32_t, int32_t, const char *, double *) { TODO("aasr_feat_run_f64"); }
aasr_status aasr_feat_run_batch_dev(aasr_feat *, const int16_t *, const int64_t *, const int64_t *, int32_t, float *, void *) { TODO("aasr_feat_run_batch_dev"); }
aasr_status aasr_feat_set_parameters(aasr_feat *, const char *, const char *) { TODO("aasr_feat_set_parameters"); }
aasr_status aasr_recipe_batch_range(int32_t, int32_t, int32_t, int32_t *, int32_t *) { TODO("aasr_recipe_batch_range"); }
aasr_status aasr_run_recipe(aasr_feat *, aasr_gmm *, const char *, const aasr_run_options *, aasr_run_stats *) { TODO("aasr_run_recipe"); }
aasr_status aasr_run_utterance(aasr_feat *, aasr_gmm *, const int16_t *, int64_t, int32_t, int32_t, int, int, uint8_t **, int64_t *, int64_t *) { TODO("aasr_run_utterance"); }
}
