// Shared by the translation units of libffn_hip.so (not part of the C-ABI).
#ifndef FFN_INTERNAL_H_
#define FFN_INTERNAL_H_

// Stores a printf-formatted message for ffn_last_error() (thread local) and
// returns `code`.  Defined in ffn_hip.hip.
int ffn_set_error(int code, const char* fmt, ...)
    __attribute__((format(printf, 2, 3)));

// What other translation units may know about a device canvas.
struct FfnCanvasView {
  int device_id;
  void* engine_stream;  // hipStream_t of the owning engine
  const float* image;            // f32 canvas image, or NULL for a uint8 canvas:
  const unsigned char* image_u8;  //   raw image ...
  const float* image_lut;         //   ... and its 256-entry normalisation table
  const int* segmentation;
  long long shape_zyx[3];
};
struct ffn_canvas;
int ffn_canvas_view(ffn_canvas* canvas, FfnCanvasView* out);

#endif  // FFN_INTERNAL_H_
