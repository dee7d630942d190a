// p3_bucket_reader.h -- streaming reader for one partition bucket (all of its (bucket, workgroup) segments).
//
// The scatter kernels leave bucket pb as P3_NWG segments that are contiguous in memory, each padded to a
// multiple of 16 records (k_part_scan).  Walking segment by segment makes every wave wait on a chain of
// dependent loads (offset -> count -> records) per ~500 records.  Instead the bucket is read as ONE flat,
// fully coalesced range of 16-record chunks; a per-chunk "valid records" byte table in LDS (built once per
// workgroup from the 256 segment descriptors) tells padding from data.  Buckets larger than the table are
// processed in windows.
#pragma once
// (included by p3_kernels_part.h after the partition constants are defined)

#define P3_VALID_CAP 24576u  // chunks per window (bytes of LDS) = 393,216 records

// Calls f(rec) for every real record of bucket pb.  ALL threads of the workgroup must call this (it
// synchronises).  valid: LDS byte array of P3_VALID_CAP entries; sc2: two LDS u32 scratch words.
template <typename REC, typename F>
__device__ __forceinline__ void bucket_for_each(const REC* __restrict__ part, const u32* __restrict__ seg_off,
                                                const u32* __restrict__ ghist, u32 pb, uint8_t* valid, u32* sc2,
                                                F f) {
  constexpr u32 RPV = 16 / sizeof(REC);  // records per 16-byte vector
  constexpr u32 VPC = 16 / RPV;          // vectors per 16-record chunk
  // segment descriptor of thread t < NWG
  u32 my_off = 0, my_n = 0;
  if (threadIdx.x < P3_NWG) {
    my_off = seg_off[pb * P3_NWG + threadIdx.x];
    my_n = ghist[pb * P3_NWG + threadIdx.x];
  }
  if (threadIdx.x == 0) sc2[0] = my_off;
  if (threadIdx.x == P3_NWG - 1) sc2[1] = my_off + ((my_n + 15u) & ~15u);
  __syncthreads();
  const u32 start = sc2[0];                    // first record of the bucket (multiple of 16)
  const u32 nch = (sc2[1] - start) >> 4;       // chunks in the bucket
  const u32 my_c0 = (my_off - start) >> 4;     // first chunk of my segment
  const u32 my_nc = (my_n + 15u) >> 4;         // chunks of my segment
  for (u32 wb = 0; wb < nch; wb += P3_VALID_CAP) {
    const u32 wn = (nch - wb) < P3_VALID_CAP ? (nch - wb) : P3_VALID_CAP;
    if (threadIdx.x < P3_NWG) {
      // chunks [my_c0, my_c0+my_nc) intersected with [wb, wb+wn)
      u32 lo = my_c0 > wb ? my_c0 : wb;
      u32 hi = (my_c0 + my_nc) < (wb + wn) ? (my_c0 + my_nc) : (wb + wn);
      for (u32 c = lo; c < hi; c++) {
        const u32 left = my_n - ((c - my_c0) << 4);
        valid[c - wb] = (uint8_t)(left < 16u ? left : 16u);
      }
    }
    __syncthreads();
    const uint4* vec = reinterpret_cast<const uint4*>(part + start) + (u64)wb * VPC;
    const u32 nvec = wn * VPC;
    // four 16-byte loads in flight per lane before any record is consumed (the consumers are LDS-latency
    // chains; without this a wave has a single outstanding HBM request)
    auto consume = [&](const uint4& v, u32 j) {
      const u32 nvalid = valid[j / VPC];
      const u32 first = (j % VPC) * RPV;  // index of this vector's first record inside its chunk
      if constexpr (sizeof(REC) == 4) {
        const u32 r4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (u32 t = 0; t < 4; t++)
          if (first + t < nvalid) f(r4[t]);
      } else if constexpr (sizeof(REC) == 8) {
        const u64 r2[2] = {((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z};
#pragma unroll
        for (u32 t = 0; t < 2; t++)
          if (first + t < nvalid) f(r2[t]);
      } else {
        REC r;
        r.lo = ((u64)v.y << 32) | v.x;
        r.hi = ((u64)v.w << 32) | v.z;
        if (first < nvalid) f(r);
      }
    };
    const u32 B = blockDim.x;
    u32 j = threadIdx.x;
    for (; j + 3 * B < nvec; j += 4 * B) {
      const uint4 v0 = vec[j], v1 = vec[j + B], v2 = vec[j + 2 * B], v3 = vec[j + 3 * B];
      consume(v0, j);
      consume(v1, j + B);
      consume(v2, j + 2 * B);
      consume(v3, j + 3 * B);
    }
    for (; j < nvec; j += B) {
      const uint4 v = vec[j];
      consume(v, j);
    }
    __syncthreads();
  }
}
