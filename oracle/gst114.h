/* oracle/gst114.h — TEST INFRASTRUCTURE ONLY.  See gst114.c. */
#ifndef GST114_ORACLE_H
#define GST114_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
enum { GST114_BT601 = 0, GST114_BT709 = 1, GST114_BT2020 = 2 };
enum { GST114_BGRA = 0, GST114_RGBA = 1 };
enum { GST114_BILINEAR = 0, GST114_NEAREST = 1, GST114_BICUBIC = 2 };   /* BICUBIC = videoscale method=catrom */

int gst114_set_threads (int n);
void gst114_yuv_to_rgb (int matrix, int Y, int U, int V, int *r, int *g, int *b);
int gst114_yuv420_to_rgb (const uint8_t *yp, int ys, const uint8_t *up, int us, const uint8_t *vp, int vs,
    int planar, int w, int h, int matrix, int cosited, int out_format, uint8_t *out, int os);
void gst114_linear_taps (int in, int out, int j, int prec, int *i0, int *i1, int *t0, int *t1);
void gst114_vtaps (int in_h, int out_h, int y, int *i0, int *i1, int *w);
uint32_t gst114_hinc (int in_w, int out_w);
int gst114_nearest_index (int in, int out, int j);
/* n-tap set-up of GstVideoResampler (cubic, b = 0, c = 0.5), 6-bit taps: returns n_taps; idx / taps: out * n_taps entries */
int gst114_cubic_taps (int in, int out, int *idx, int *taps, int max_entries);
int gst114_scale_4u8 (const uint8_t *in, int is, int w, int h, uint8_t *out, int os, int ow, int oh, int method);
int gst114_rgb_to_yuv420 (const uint8_t *in, int is, int in_format, int w, int h, int matrix, int cosited,
    int planar, uint8_t *yp, int ys, uint8_t *up, int us, uint8_t *vp, int vs);
int gst114_scale_plane (const uint8_t *in, int is, int w, int h, int n, uint8_t *out, int os, int ow, int oh);
int gst114_packed422_to_rgb (const uint8_t *in, int is, int yuy2, int w, int h, int matrix, int cosited, int out_format, uint8_t *out, int os);
int gst114_convertscale_packed422 (const uint8_t *in, int is, int yuy2, int w, int h, int matrix, int cosited, int out_format, int method,
    uint8_t *out, int os, int ow, int oh);
/* packed 4:2:2 outputs / packed -> 4:2:0 (see gst114.c) */
int gst114_rgb_to_packed422 (const uint8_t *in, int is, int in_format, int w, int h, int matrix, int cosited, int yuy2, uint8_t *out, int os);
int gst114_yuv420_to_packed422 (const uint8_t *yp, int ys, const uint8_t *up, int us, const uint8_t *vp, int vs,
    int planar, int w, int h, int cosited_in, int cosited_out, int yuy2, uint8_t *out, int os);
int gst114_packed422_swizzle (const uint8_t *in, int is, int in_yuy2, int w, int h, int out_yuy2, uint8_t *out, int os);
int gst114_packed422_to_yuv420 (const uint8_t *in, int is, int yuy2, int w, int h, int cosited_in, int cosited_out, int planar,
    uint8_t *yp, int ys, uint8_t *up, int us, uint8_t *vp, int vs);
int gst114_scale_packed422 (const uint8_t *in, int is, int yuy2, int w, int h, uint8_t *out, int os, int ow, int oh);
int gst114_scale_plane_cubic (const uint8_t *in, int is, int w, int h, int n, uint8_t *out, int os, int ow, int oh, int chroma);
int gst114_linear_ntaps (int in, int out, int *idx, int *taps, int max_entries);
int gst114_scale_packed422_cubic (const uint8_t *in, int is, int yuy2, int w, int h, uint8_t *out, int os, int ow, int oh);
int gst114_scale_plane_nearest (const uint8_t *in, int is, int w, int h, int n, uint8_t *out, int os, int ow, int oh);
int gst114_scale_packed422_nearest (const uint8_t *in, int is, int yuy2, int w, int h, uint8_t *out, int os, int ow, int oh);
/* videoconvert YUV -> YUV at one size when the matrix and / or the chroma siting change (sample addressing: gst114.c) */
int gst114_yuv_to_yuv (const uint8_t *yp, int ys, int ystep, const uint8_t *up, const uint8_t *vp, int cs, int cstep, int in420, int w, int h,
    int matrix_in, int cosited_in, int matrix_out, int cosited_out,
    uint8_t *oy, int oys, int oystep, uint8_t *ou, uint8_t *ov, int ocs, int ocstep, int out420);
int gst114_default_matrix (int height);
int gst114_default_cosited (int height);
int gst114_convertscale_yuv420 (const uint8_t *yp, int ys, const uint8_t *up, int us, const uint8_t *vp, int vs,
    int planar, int w, int h, int matrix, int cosited, int out_format, int method,
    uint8_t *out, int os, int ow, int oh);
#ifdef __cplusplus
}
#endif
#endif
