// Host-side work either side of the training step, moved to the device:
//   negative sampling  -- sampler/sampler.py:16-48: per user, num_negatives items drawn uniformly, redrawn while
//                         (user, item) is an observed pair.  The reference walks a Python set in a double loop
//                         (943 x 30 draws: ~0.1 s; the scripts call it three times per run); here one thread per
//                         (user, draw) tests a bitmap of the observed pairs.
//   feature assembly   -- data/reader.py:98-101 feature(): [user_id, item_id] joined with the user row (age,
//                         gender one-hot, occupation one-hot) and the item row (19 genre flags) into the (B,45)
//                         float matrix every feature model reads.  Two pandas merges there, one gather here.
// Integer / byte work: HBM-bound, bit-exact (no floating-point arithmetic besides int -> float of the ids).
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// thread t = user * num_neg + j.  Draw k of that slot is item = floor(u32 * num_items / 2^32), u32 from a
// counter-based hash of (seed, t, k): no state, any launch geometry gives the same sample.
__global__ void __launch_bounds__(kBlock)
negative_sample_kernel(const uint32_t* __restrict__ excluded, int64_t words_per_user, int64_t num_users, int64_t num_items,
                       int num_neg, uint64_t seed, int max_tries, int64_t* __restrict__ users, int64_t* __restrict__ items,
                       int32_t* __restrict__ fail) {
  const int64_t total = num_users * num_neg;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = t / num_neg;
    const uint32_t* row = excluded + u * words_per_user;
    int64_t item = 0;
    bool found = false;
    for (int k = 0; k < max_tries; ++k) {
      const uint64_t h = mix64(seed ^ mix64((uint64_t)t * 0x100000001B3ull + (uint64_t)k));
      item = (int64_t)(((h >> 32) * (uint64_t)num_items) >> 32);
      if (((row[item >> 5] >> (item & 31)) & 1u) == 0u) {
        found = true;
        break;
      }
    }
    if (!found && fail) *fail = 1;  // (practically) every item of this user is an observed pair
    users[t] = u;
    items[t] = item;
  }
}

__global__ void __launch_bounds__(kBlock)
assemble_kernel(const int64_t* __restrict__ users, const int64_t* __restrict__ items, int64_t n,
                const float* __restrict__ ufeat, int uw, int64_t nu, const float* __restrict__ ifeat, int iw, int64_t ni,
                float* __restrict__ out, int64_t ldo, int32_t* __restrict__ err) {
  const int width = 2 + uw + iw;
  const int64_t total = n * width;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = g / width;
    const int c = (int)(g - b * width);
    int64_t u = users[b], i = items[b];
    if (u < 0 || u >= nu || i < 0 || i >= ni) {
      if (err) *err = 1;
      u = u < 0 || u >= nu ? 0 : u;
      i = i < 0 || i >= ni ? 0 : i;
    }
    float v;
    if (c == 0) v = (float)users[b];
    else if (c == 1) v = (float)items[b];
    else if (c < 2 + uw) v = ufeat[u * uw + (c - 2)];
    else v = ifeat[i * iw + (c - 2 - uw)];
    out[b * ldo + c] = v;
  }
}

}  // namespace

extern "C" int ctr_negative_sample(const uint32_t* excluded, int64_t words_per_user, int64_t num_users, int64_t num_items,
                                   int num_negatives, uint64_t seed, int64_t* users, int64_t* items, int32_t* fail_flag,
                                   void* stream) {
  CTR_REQUIRE(num_users >= 0 && num_negatives >= 0, CTR_EINVAL);
  if (num_users == 0 || num_negatives == 0) return CTR_OK;
  CTR_REQUIRE(excluded && users && items && num_items >= 1 && num_items < (1ll << 32) &&
                  words_per_user * 32 >= num_items,
              CTR_EINVAL);
  const int grid = ctr_stream_grid(num_users * num_negatives, kBlock);
  hipLaunchKernelGGL(negative_sample_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, excluded, words_per_user,
                     num_users, num_items, num_negatives, seed, 1 << 14, users, items, fail_flag);
  return ctr_launch_status();
}

extern "C" int ctr_assemble_features(const int64_t* users, const int64_t* items, int64_t n, const float* user_feat,
                                     int user_width, int64_t num_users, const float* item_feat, int item_width,
                                     int64_t num_items, float* out, int64_t ldo, int32_t* err_flag, void* stream) {
  CTR_REQUIRE(n >= 0, CTR_EINVAL);
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(users && items && user_feat && item_feat && out && user_width >= 0 && item_width >= 0 && num_users > 0 &&
                  num_items > 0 && ldo >= 2 + user_width + item_width,
              CTR_EINVAL);
  const int grid = ctr_stream_grid(n * (2 + user_width + item_width), kBlock);
  hipLaunchKernelGGL(assemble_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, users, items, n, user_feat,
                     user_width, num_users, item_feat, item_width, num_items, out, ldo, err_flag);
  return ctr_launch_status();
}
