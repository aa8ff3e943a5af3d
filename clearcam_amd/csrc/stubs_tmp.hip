// TEMPORARY: not-yet-implemented entry points (replaced by clip.hip / index.hip).
#include "common.h"
#include "../../include/clearcam_hip.h"
#define NI { cc::set_error("not implemented yet"); return -38; }
extern "C" {
int cc_clip_create(cc_clip**, const cc_clip_config*, int, int) NI
int cc_clip_load(cc_clip*, const char*, const float*, const int64_t*, int) NI
int cc_clip_finalize(cc_clip*) NI
int cc_clip_encode_image(cc_clip*, const float*, int, int, float*, int, void*) NI
int cc_clip_encode_text(cc_clip*, const int32_t*, int, float*, int, void*) NI
int cc_clip_last_gpu_ms(cc_clip*, float*) NI
void cc_clip_destroy(cc_clip*) {}
int cc_index_create(cc_index**, int, int64_t, int) NI
int cc_index_add(cc_index*, const float*, int64_t, int) NI
int cc_index_size(cc_index*, int64_t*) NI
int cc_index_scores(cc_index*, const float*, int, float*, int, void*) NI
int cc_index_search(cc_index*, const float*, int, int, int32_t*, float*, int, void*) NI
void cc_index_destroy(cc_index*) {}
}
