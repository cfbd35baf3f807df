/* vitcap_jpeg.h -- C ABI of the JPEG front half on the host (vitcap_amd/libvitcap_jpeg.so, built from vitcap_amd/csrc/jpeg_host.cpp by
 * g++: no HIP, no torch -- the loader's worker processes load it through ctypes next to numpy only).
 *
 * What it replaces.  The reference decodes a (key, base64 JPEG) TSV row with cv2.imdecode inside DataLoader workers
 * (src/data_layer/transform.py:106-136 ImageTransform2Dict -> src/tools/common.py:23-31 img_from_base64) and runs the torchvision
 * transforms on the host; SURVEY.md section 8f row 1 puts that split where a GPU wants it: ENTROPY decoding (sequential, branchy) stays
 * on the host, everything behind it -- dequantisation, 8x8 inverse DCT, chroma upsampling, YCbCr -> RGB, then the resize / crop /
 * normalise of csrc/preproc.hip -- runs on the device (csrc/jpeg.hip: vitcap_jpeg_backhalf in vitcap_hip.h).
 *
 * Parity: the device back half restates libjpeg(-turbo)'s DEFAULT decompression path -- jidctint.c jpeg_idct_islow, jdsample.c
 * h2v1 / h2v2 "fancy" upsampling, jdcolor.c ycc_rgb_convert -- which is integer arithmetic end to end, so the bar is BIT-EXACT pixels
 * against Pillow (which wraps libjpeg-turbo with those defaults; cv2.imdecode wraps the same library and defaults, unverified here:
 * cv2 is not installed in this image).  Streams outside the supported subset are REFUSED (VITCAP_JPEG_EUNSUPPORTED) and the caller
 * decodes them with Pillow: progressive / arithmetic / lossless / 12-bit, CMYK or RGB colour spaces, more than one scan, sampling
 * factors other than 1x1 (4:4:4 / grey), 2x1 (4:2:2) or 2x2 (4:2:0) luma with 1x1 chroma, chroma planes narrower than 3 samples.
 */
#ifndef VITCAP_JPEG_H
#define VITCAP_JPEG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VITCAP_JPEG_ABI 1
#define VITCAP_JPEG_OK 0
#define VITCAP_JPEG_EINVAL 1        /* not a JPEG / truncated / corrupt */
#define VITCAP_JPEG_EUNSUPPORTED 2  /* a valid stream outside the subset above: decode it with Pillow */

/* Geometry and tables of one image (host memory; the same struct is handed to the device back half).  Component c covers
 * blocks_w[c] x blocks_h[c] blocks of 8x8 samples (whole MCUs: the padded size), of which samp_w[c] x samp_h[c] samples are real
 * (libjpeg's downsampled_width / downsampled_height).  Coefficient blocks of component c start at block index block0[c] of the image's
 * coefficient array: [block row][block column][64] int16, NATURAL (row-major) order inside a block, NOT dequantised. */
typedef struct vitcap_jpeg_info {
  int32_t abi;
  int32_t width, height;          /* image size in pixels */
  int32_t ncomp;                  /* 1 (grey) or 3 (YCbCr) */
  int32_t hs[3], vs[3];           /* sampling factors */
  int32_t blocks_w[3], blocks_h[3];
  int32_t samp_w[3], samp_h[3];
  int32_t block0[3];
  int32_t nblocks;                /* all components */
  uint16_t qt[3][64];             /* quantisation table of each component, natural order */
} vitcap_jpeg_info;

int vitcap_jpeg_abi(void);
/* Reads the headers up to the first scan.  VITCAP_JPEG_OK: *info is filled and the stream is inside the supported subset. */
int vitcap_jpeg_parse(const uint8_t* data, size_t n, vitcap_jpeg_info* info);
/* Entropy-decodes the scan into coefs[info->nblocks * 64] (zero-filled here).  `info` must come from vitcap_jpeg_parse of the same bytes. */
int vitcap_jpeg_decode_coefs(const uint8_t* data, size_t n, const vitcap_jpeg_info* info, int16_t* coefs);
const char* vitcap_jpeg_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
