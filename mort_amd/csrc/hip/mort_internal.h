/*
 * mort_internal.h -- functions shared between the translation units of libmort_hip.so (not part of the C ABI).
 */
#ifndef MORT_INTERNAL_H
#define MORT_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>

/* tile_sort.hip */
size_t mort_tile_sort_temp_bytes(int n);
hipError_t mort_tile_sort_desc(const unsigned *d_cost, unsigned *d_keys_out, unsigned *d_iota, unsigned *d_order, void *d_temp,
                               size_t temp_bytes, int n, hipStream_t s);
hipError_t mort_tile_heavy_count(const unsigned *d_keys_desc, int n, unsigned percent, unsigned max_r, const unsigned long long *d_frame_total,
                                 unsigned long long lanes, unsigned *d_out, hipStream_t s);

#endif
