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

#define P3_VALID_CAP 8192u  // chunks per window (bytes of LDS) = 131,072 records

// Calls f(recs, active_mask) with batches of 4*(16/sizeof(REC)) records of bucket pb (bit i of the mask says
// whether recs[i] is a real record).  ALL threads of the workgroup must call this (it
// synchronises).  valid: LDS byte array of P3_VALID_CAP entries; sc2: two LDS u32 scratch words.
// TAGGED: the table entries are 16 bits wide (so the window is VCAP / 2 chunks): the high byte carries
// (segment index / tag_div) and f gets the tag of each of the four vectors as a third argument (k_part_count: the
// source -- rank or position chunk -- of a re-partitioned cell's segment group).
// INDEXED: f gets, as its last argument, the index in `part` of the record recs[0] came from -- recs[i] came from
// part[that + (i / RPV) * blockDim.x * RPV + i % RPV] (k_kmer_dedup16 queues a record it cannot place at once by index).
template <typename REC, u32 VCAP = P3_VALID_CAP, bool TAGGED = false, bool INDEXED = false, typename F>
__device__ __forceinline__ void bucket_for_each(const REC* __restrict__ part, const u32* __restrict__ seg_off,
                                                const u32* __restrict__ ghist, u32 pb, uint8_t* valid, u32* sc2,
                                                F f, u32 seg_begin = 0, u32 seg_count = P3_NWG, u32 tag_div = P3_NWG) {
  constexpr u32 RPV = 16 / sizeof(REC);  // records per 16-byte vector
  constexpr u32 VC = TAGGED ? VCAP / 2 : VCAP;  // chunks per window
  unsigned short* valid16 = reinterpret_cast<unsigned short*>(valid);
  constexpr u32 VPC = 16 / RPV;          // vectors per 16-record chunk
  // segment descriptor of thread t < seg_count (segments [seg_begin, seg_begin+seg_count) of row pb)
  u32 my_off = 0, my_n = 0;
  if (threadIdx.x < seg_count) {
    my_off = seg_off[pb * P3_NWG + seg_begin + threadIdx.x];
    my_n = ghist[pb * P3_NWG + seg_begin + threadIdx.x];
  }
  __syncthreads();  // sc2 / valid may still be in use by a previous call
  if (threadIdx.x == 0) sc2[0] = my_off;
  if (threadIdx.x == seg_count - 1) sc2[1] = my_off + ((my_n + 15u) & ~15u);
  __syncthreads();
  const u32 start = sc2[0];                    // first record of the bucket (multiple of 16)
  const u32 nch = (sc2[1] - start) >> 4;       // chunks in the bucket
  const u32 my_c0 = (my_off - start) >> 4;     // first chunk of my segment
  const u32 my_nc = (my_n + 15u) >> 4;         // chunks of my segment
  for (u32 wb = 0; wb < nch; wb += VC) {
    const u32 wn = (nch - wb) < VC ? (nch - wb) : VC;
    // segments need not abut (estimated-capacity partitions leave gaps): unmarked chunks hold no records
    for (u32 i = threadIdx.x; i < ((TAGGED ? 2 * wn : wn) + 3) / 4; i += blockDim.x) reinterpret_cast<u32*>(valid)[i] = 0;
    __syncthreads();
    if (threadIdx.x < seg_count) {
      // chunks [my_c0, my_c0+my_nc) intersected with [wb, wb+wn)
      u32 lo = my_c0 > wb ? my_c0 : wb;
      u32 hi = (my_c0 + my_nc) < (wb + wn) ? (my_c0 + my_nc) : (wb + wn);
      for (u32 c = lo; c < hi; c++) {
        const u32 left = my_n - ((c - my_c0) << 4);
        const u32 e = left < 16u ? left : 16u;
        if constexpr (TAGGED) valid16[c - wb] = (unsigned short)(e | ((threadIdx.x / tag_div) << 8));
        else valid[c - wb] = (uint8_t)e;
      }
    }
    __syncthreads();
    const uint4* vec = reinterpret_cast<const uint4*>(part + start) + (u64)wb * VPC;
    const u32 nvec = wn * VPC;
    // Four 16-byte loads in flight per lane; the 4*RPV records they hold are handed to f as ONE batch so
    // that the consumer can issue its LDS lookups for all of them before using any (LDS latency chains,
    // not HBM, bound these kernels at 4 waves per SIMD).
    const u32 B = blockDim.x;
    // all four table bytes first (one LDS round trip, clamped index instead of a branch), then the four loads
    // back to back; a vector is fetched only if its chunk holds a record at or beyond the vector's first slot
    // (gaps between estimated-capacity segments and segment tails cost no memory traffic): a lane with nothing
    // to fetch re-reads the window's first vector (one broadcast line) and its records stay masked out by `act`
    auto fetch = [&](u32 jb, uint4 (&v)[4], u32 (&nval)[4]) {  // nval: table byte (count, and tag if TAGGED)
      const u32 j0 = jb + threadIdx.x;
#pragma unroll
      for (u32 u = 0; u < 4; u++) {
        const u32 ci = (j0 + u * B) / VPC;
        if constexpr (TAGGED) nval[u] = (u32)valid16[ci < VC ? ci : VC - 1];
        else nval[u] = (u32)valid[ci < VC ? ci : VC - 1];
      }
#pragma unroll
      for (u32 u = 0; u < 4; u++) {
        const u32 j = j0 + u * B;
        if (j >= nvec) nval[u] = 0;
        v[u] = vec[((nval[u] & 31u) > (j % VPC) * RPV) ? j : 0u];
      }
    };
    // software pipeline: the loads of batch i+1 are in flight while the consumer works on batch i
    uint4 vn[4];
    u32 nvn[4];
    if (nvec) fetch(0, vn, nvn);
    for (u32 jb = 0; jb < nvec; jb += 4 * B) {  // uniform trip count: consumers use wave-wide ballots
      const u32 j0 = jb + threadIdx.x;
      uint4 v[4];
      u32 nval[4];
#pragma unroll
      for (u32 u = 0; u < 4; u++) {
        v[u] = vn[u];
        nval[u] = nvn[u];
      }
      if (jb + 4 * B < nvec) fetch(jb + 4 * B, vn, nvn);
      REC recs[4 * RPV];
      u32 act = 0;
      u32 tags[4];
#pragma unroll
      for (u32 u = 0; u < 4; u++) {
        const u32 nvalid = nval[u] & 31u;
        tags[u] = nval[u] >> 8;
        const u32 first = ((j0 + u * B) % VPC) * RPV;
        if constexpr (sizeof(REC) == 2) {
          const u32 w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (u32 t = 0; t < 4; t++) {
            recs[8 * u + 2 * t] = (REC)(w4[t] & 0xffffu);
            recs[8 * u + 2 * t + 1] = (REC)(w4[t] >> 16);
          }
        } else if constexpr (sizeof(REC) == 4) {
          recs[4 * u + 0] = v[u].x;
          recs[4 * u + 1] = v[u].y;
          recs[4 * u + 2] = v[u].z;
          recs[4 * u + 3] = v[u].w;
        } else if constexpr (sizeof(REC) == 8) {
          recs[2 * u + 0] = ((u64)v[u].y << 32) | v[u].x;
          recs[2 * u + 1] = ((u64)v[u].w << 32) | v[u].z;
        } else {
          recs[u].lo = ((u64)v[u].y << 32) | v[u].x;
          recs[u].hi = ((u64)v[u].w << 32) | v[u].z;
        }
#pragma unroll
        for (u32 t = 0; t < RPV; t++)
          if (first + t < nvalid) act |= 1u << (u * RPV + t);
      }
      if constexpr (TAGGED) f(recs, act, tags);
      else if constexpr (INDEXED) f(recs, act, (u32)(start + ((u64)wb * VPC + j0) * RPV));
      else f(recs, act);
    }
    __syncthreads();
  }
}
