// svo_track.hip - Tracking::Track (reference src/Tracking.cc:180-252) on the device.
#include "svo_internal.h"

extern "C" int svo_track_reset(svo_ctx* ctx, const svo_camera* cam) {
  if (!ctx || !cam) return SVO_E_INVALID;
  ctx->cam = *cam;
  ctx->track_frame = 0;
  return SVO_OK;
}
extern "C" int svo_track_frame(svo_ctx*, const uint8_t*, int, const uint8_t*, int, double,
                               const int32_t*, int, svo_track_result*) {
  return SVO_E_INVALID;
}
extern "C" int svo_track_batch_dev(svo_ctx*, const uint8_t*, const uint8_t*, int, int,
                                   svo_track_result*) {
  return SVO_E_INVALID;
}
