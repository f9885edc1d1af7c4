/*
 * libffn_hip.so, label operations -- the integer side of subvolume assembly
 * ("overlapping-subvolume reconciliation / global ID
 * assembly").  HBM-bound streaming kernels over label volumes; all results are
 * bit-exact integers.  Reference interfaces taken over (file:line relative to
 * the google/ffn checkout):
 *
 *   ffn_labels_pair_counts + ffn_labels_apply_pair_labels
 *       np.unique(a | b << 32, return_inverse, return_counts) and the final
 *       gather of segmentation.split_segmentation_by_intersection
 *       (ffn/inference/segmentation.py:181-290, the np.unique at :259-260 and
 *       the relabel at :290); also the overlap-zone (id_a, id_b, count) table of
 *       the union-find assembly described in doc/manual.md:119-127.
 *   ffn_labels_remap
 *       lookup-table relabelling: clear_dust (segmentation.py:21-63),
 *       make_labels_contiguous-style maps (inference.py:709), global id
 *       offsets + union-find roots when assembling sub-boxes.
 *   ffn_labels_connected_components
 *       connectomics.segmentation.labels.split_disconnected_components as
 *       called by segmentation.clean_up_and_count (segmentation.py:161-162):
 *       components of equal non-zero label, numbered 1.. in raster order of
 *       their first voxel (skimage.measure.label order), 0 stays 0.
 *
 * Conventions as in ffn_hip.h: plain C types, 0 / negative FFN_ERR_* return
 * codes, ffn_last_error() for the message, caller owns host buffers.  Label
 * volumes are flat C-order arrays of `elem_bytes` = 4 (uint32 / int32 bit
 * pattern) or 8 (uint64) bytes per voxel.  A handle owns one HIP stream and
 * grow-only device scratch; calls on one handle must be serialised.
 */
#ifndef FFN_LABELS_H_
#define FFN_LABELS_H_

#include <stddef.h>
#include <stdint.h>

#include "ffn_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ffn_labels ffn_labels;

int ffn_labels_create(int device_id, ffn_labels** out);
void ffn_labels_destroy(ffn_labels* h);

/* Joint histogram of the pairs (a[i], b[i]), i < n.  b == NULL counts a alone
 * (pair_b is then all 0).  Every id must be < 2^32 - 1 (the reference remaps
 * larger ids first, segmentation.py:208-243; so does the Python caller).
 * Writes up to `cap` unique pairs in UNSPECIFIED order (the caller sorts);
 * *n_pairs is the true number of unique pairs (FFN_ERR_ARG if > cap).
 * pair_slot[k] identifies pair k for ffn_labels_apply_pair_labels.  The uploaded
 * volumes stay resident on the device until the next call on this handle. */
int ffn_labels_pair_counts(ffn_labels* h, const void* a, const void* b,
                           int elem_bytes, size_t n, size_t cap,
                           uint64_t* pair_a, uint64_t* pair_b,
                           uint64_t* pair_count, uint32_t* pair_slot,
                           size_t* n_pairs);

/* out[i] = new_label[k] where k is the pair of voxel i in the volumes of the
 * preceding ffn_labels_pair_counts call on this handle (same n, elem_bytes). */
int ffn_labels_apply_pair_labels(ffn_labels* h, size_t n_pairs,
                                 const uint32_t* pair_slot,
                                 const uint64_t* new_label, void* out);

/* out[i] = values[j] if in[i] == keys[j] for some j, else (keep_missing ?
 * in[i] : 0).  keys must be unique and != 2^64 - 1.  in == out is allowed. */
int ffn_labels_remap(ffn_labels* h, const void* in, int elem_bytes, size_t n,
                     size_t n_keys, const uint64_t* keys,
                     const uint64_t* values, int keep_missing, void* out);

/* Connected components of equal non-zero label.  connectivity 1 / 2 / 3 = 6 /
 * 18 / 26 neighbours.  out (same elem_bytes as in) gets ids 1..*n_components in
 * raster order of each component's first voxel; 0 stays 0.  Optional outputs
 * (NULL to skip), for up to `cap` components: first_index[k] = flat index of the
 * first voxel of component k+1, sizes[k] = its voxel count.  *first_zero_index =
 * flat index of the first 0 voxel, or -1.  n = prod(shape) must be < 2^32 - 1. */
int ffn_labels_connected_components(ffn_labels* h, const void* in,
                                    int elem_bytes, const int64_t shape_zyx[3],
                                    int connectivity, void* out,
                                    uint64_t* n_components, size_t cap,
                                    uint64_t* first_index, uint64_t* sizes,
                                    int64_t* first_zero_index);

/* ---- device-resident assembly (the final segmentation merge) ----
 * The same operations on DEVICE pointers (int32 labels, e.g. a canvas'
 * segmentation from ffn_canvas_view or the data_ptr of the tensor RCCL reduces),
 * so that assembling the sub-boxes of a volume moves no voxel through the
 * host.  Every call returns with its kernels complete (stream synchronised);
 * the caller synchronises whatever produced the inputs. */

/* dst[i] = src[i] > 0 ? src[i] : 0 (a canvas' -1 "excluded" markers dropped). */
int ffn_labels_copy_device(ffn_labels* h, const int32_t* src_dev, size_t n,
                           int32_t* dst_dev);

/* The same from a live device canvas (Canvas.segmentation, inference.py:224-232)
 * straight out of its HBM: dst_dev holds cz * cy * cx int32. */
int ffn_labels_copy_canvas(ffn_labels* h, ffn_canvas* canvas, int32_t* dst_dev);

/* Writes the CORE [core_lo, core_hi) (sub-box coordinates) of a sub-box's
 * labels, globally offset (v > 0 ? v + id_offset : 0), into the assembled
 * volume, in which the sub-box sits at corner_zyx
 * (ffn/utils/bounding_box.py:250-412 tiling; doc/manual.md:107-127). */
int ffn_labels_place_core_device(ffn_labels* h, const int32_t* src_dev,
                                 const int64_t src_shape_zyx[3],
                                 const int64_t core_lo[3], const int64_t core_hi[3],
                                 int32_t id_offset, int32_t* dst_dev,
                                 const int64_t dst_shape_zyx[3],
                                 const int64_t corner_zyx[3]);

/* Joint histogram over a sub-box's MARGIN (everything outside its core) of
 * (own label + id_offset, assembled label): the overlap table of the
 * union-find reconciliation (doc/manual.md:119-127).  Output as
 * ffn_labels_pair_counts (pairs with a 0 member included; unsorted). */
int ffn_labels_margin_pairs_device(ffn_labels* h, const int32_t* own_dev,
                                   const int64_t own_shape_zyx[3],
                                   int32_t id_offset, const int64_t core_lo[3],
                                   const int64_t core_hi[3],
                                   const int32_t* assembled_dev,
                                   const int64_t assembled_shape_zyx[3],
                                   const int64_t corner_zyx[3], size_t cap,
                                   uint64_t* pair_a, uint64_t* pair_b,
                                   uint64_t* pair_count, size_t* n_pairs);

/* In place: vol[i] = values[j] where vol[i] == keys[j], unchanged otherwise. */
int ffn_labels_remap_device(ffn_labels* h, int32_t* vol_dev, size_t n,
                            size_t n_keys, const uint64_t* keys,
                            const uint64_t* values);

/* HIP-event time of the kernels (no host<->device copies) of the last call on
 * this handle, and the HBM bytes they are specified to move (algorithmic):
 * the measurement hook for the HBM roofline of these kernels. */
int ffn_labels_last_timing(ffn_labels* h, double* kernel_ms,
                           double* algorithmic_bytes);

#ifdef __cplusplus
}
#endif
#endif /* FFN_LABELS_H_ */
