// Photon propagator for gfx950 (MI355X), pooled scheduling.
//
// Same arithmetic as prop_kernel.hip (every device function is shared, prop_device.hip.h) and the same work
// units (slices of steps from eight sub-queues, one RNG stream per step handed on through 64-byte work records);
// what differs is how a wave keeps its 64 lanes busy.
//
// The classic kernel holds exactly one photon per lane.  A lane whose photon has died waits until k_new = 12 lanes of
// the wave are in the same position, then the wave runs photon creation (~900 instructions) for those 12 lanes: 16 %
// of all issued instructions at 19 % lane use, and 11 % of the lane trips spent waiting for the batch.
//
// Here a wave owns U = 64 + R work-unit slots instead of 64, the surplus living in a wave-private LDS pool:
//   * `ready` ring (R entries x 21 words): photons that have been created and wait for a free lane;
//   * `pending` list (U entries x 5 words): units whose photon has died and that need their next one created (or whose
//     predecessor slice has not been published yet).
// A lane whose photon dies hands its unit to `pending` and takes a photon from `ready` in the same loop trip (a
// "service" of ~40 instructions, run when k_pop lanes need it), so lanes do not wait for creation; creation runs when
// the ring has room for a batch, for up to 64 pending units at once and with the results going to the ring.  Waiting
// for a predecessor slice costs a pending slot, not a lane.  Everything is private to the wave: no locks, no polling
// of other waves' LDS, wave barriers only (the cross-wave mailbox experiment of round 1 lost to exactly that).
// The ring is first-in first-out so that no unit -- possibly the predecessor another wave waits for -- is starved.
//
// Large workgroups (12 waves share one table image) leave the LDS to the pools: 2 workgroups x 12 waves per CU.
// Results are bit-identical to the classic kernel for every R, k_pop and creation threshold: a unit's photons are
// still created and propagated in sequence from its own stream, whichever lanes carry them.
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>

#include "prop_device.hip.h"

namespace clsimhip {

#define CLSIMHIP_STR2(x) #x
#define CLSIMHIP_STR(x) CLSIMHIP_STR2(x)
#ifndef CLSIMHIP_POOL_BLOCK
#define CLSIMHIP_POOL_BLOCK 768                 // 12 waves per workgroup, 2 workgroups per CU
#define CLSIMHIP_POOL_WAVES 6                   // waves per SIMD the register allocation aims at (<= 80 VGPRs)
#endif
#ifndef CLSIMHIP_POOL_GROUPS
#define CLSIMHIP_POOL_GROUPS 2                  // workgroups per CU the LDS is shared between (experiments: 4 groups of 7 waves, profiles/r05/ab_seven_waves.txt)
#endif
constexpr int kPoolBlock = CLSIMHIP_POOL_BLOCK;
constexpr int kPoolWavesPerBlock = kPoolBlock / 64;
constexpr int kPoolMinWaves = CLSIMHIP_POOL_WAVES;
constexpr uint32_t kReadyWords = 20;            // a created photon + its unit: five 16-byte words (round 4; 21 single words before -- the carried layer
                                                // index, which only media without tilt use, now shares a word with the unit's flags)
constexpr uint32_t kPendWords = 4;              // a pending unit in 16 bytes (round 4; five words before): see pend_store()
#ifndef CLSIMHIP_POOL_STAGE
#define CLSIMHIP_POOL_STAGE 4                  // (round 4: 8 -> 4 frees two ring entries; C2 / C5 / benchmark.py +0.3 % / +0.3 % / +0.2 %)
#endif
constexpr int kPoolStage = CLSIMHIP_POOL_STAGE;      // hit stubs a wave stages before it writes them out (one atomic on the hit counter per flush)
constexpr uint32_t kPoolFixedWords = kPoolStage * kStubWords;           // hit stub staging (a parked lane keeps its step length in a register: round 4)
constexpr uint32_t kFlagLast = 1u << 16, kFlagWaiting = 1u << 17;         // unit flags above the slice number
constexpr int kPoolMinReady = 4;                // smallest ready ring the kernel runs with
constexpr int kPoolWorthwhileReady = 8;         // smallest ring with which it is chosen over the classic kernel

// `extra`: KEEP only -- find_collisions_keep's string mask: 64 lanes x ceil(strings / 64) words
__host__ __device__ constexpr uint32_t pool_wave_words(uint32_t R, uint32_t extra) { return (kPoolFixedWords + extra + kReadyWords * R + kPendWords * (64u + R) + 3u) & ~3u; }

// A pending unit: step index (below 2^23: a converter holds at most 6 139 850 streams), stream state, photons left in the slice (below 2^23: the
// kernel's prologue caps the slice size), flags (slice number, last, waiting: 18 bits) -- 128 bits, one ds_read_b128 / ds_write_b128.  Every
// word the ring does not need for the list is a ring entry more: 0.28 % per entry at 34 (profiles/r04/ab_ring_size.txt).
// The index field is what bounds a bunch for this kernel: kPoolIndexBits + half of the 18 flag bits fill a word, so a bunch of 2^23 steps or
// more never gets here (pool_kernel_max_steps(), Converter::pooled_for(), and the launcher below refuses it) -- the classic kernel runs it.
constexpr uint32_t kPoolIndexBits = 23, kPoolFlagBits = 18, kPoolIndexMask = (1u << kPoolIndexBits) - 1u;
static_assert(kPoolIndexBits + kPoolFlagBits / 2 == 32 && kPoolFlagBits % 2 == 0, "a pending entry's index (or count) and half of its flags share one word");
static_assert((kFlagWaiting << 1) == (1u << kPoolFlagBits), "slice number + last + waiting are the 18 flag bits");
typedef uint32_t pend_entry __attribute__((ext_vector_type(4)));
DM void pend_store(uint32_t *list, uint32_t k, uint32_t sidx, uint64_t rx, uint32_t left, uint32_t flags)
{
    pend_entry e = {sidx | (flags << 23), (uint32_t)rx, (uint32_t)(rx >> 32), left | ((flags >> 9) << 23)};
    *reinterpret_cast<pend_entry *>(list + kPendWords * k) = e;
}
DM void pend_load(const uint32_t *list, uint32_t k, uint32_t &sidx, uint64_t &rx, uint32_t &left, uint32_t &flags)
{
    const pend_entry e = *reinterpret_cast<const pend_entry *>(list + kPendWords * k);
    sidx = e.x & 0x7fffffu;
    rx = (uint64_t)e.y | ((uint64_t)e.z << 32);
    left = e.w & 0x7fffffu;
    flags = (e.x >> 23) | ((e.w >> 23) << 9);
}
__host__ __device__ constexpr uint32_t pool_keep_extra_words(uint32_t num_strings) { return 64u * ((num_strings + 63u) >> 6); }

// KEEP: without STOP_PHOTONS_ON_DETECTION (SetStopDetectedPhotons(false), the reference class's default, OpenCL.cxx:86): the search
// saves every DOM the segment enters from inside (find_collisions_keep) and the photon travels on; instantiated in a translation
// unit of its own (prop_pool_keep_kernel.hip)
template <int MED, bool TILT, bool ANISO, bool FLASHER, bool FAST, bool KEEP>
__global__ void __launch_bounds__(kPoolBlock, kPoolMinWaves) prop_pool_kernel(const KParams Pvalue)
{
    const KP P0 = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    (void)Pvalue;
    {   // stage the table image: one coalesced pass of the workgroup
        const uint32_t words = P0->table_words;
        const uint32_t *src = P0->tables;
        for (uint32_t i = threadIdx.x; i < words; i += kPoolBlock) lds_words[i] = src[i];
    }
    const uint32_t R = (uint32_t)P0->pool_ready;
    const uint32_t U = 64u + R;
    // (readfirstlane: the compiler cannot know that threadIdx.x >> 6 is the same in all lanes, and everything derived from it --
    // the sub-queue, hence every unit count below and the loop's exit -- would be treated as lane-varying)
    const uint32_t wave_in_group = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t keep_extra = KEEP ? pool_keep_extra_words((uint32_t)P0->num_strings) : 0u;
    uint32_t *wave_lds = lds_words + ((P0->table_words + 3u) & ~3u) + wave_in_group * pool_wave_words(R, keep_extra);      // (16-byte aligned)
    uint32_t *stage = wave_lds;
    uint32_t *keep_mask = wave_lds + kPoolFixedWords;                      // (KEEP only)
    uint32_t *pend = wave_lds + kPoolFixedWords + keep_extra;              // (64-word multiples before it)
    uint32_t *ready = pend + kPendWords * U;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    const uint32_t n_steps = P0->n_steps;
    uint32_t slice_photons, rounds;
    {
        const uint32_t max_photons = P0->queue[1];                 // scan_steps_kernel
        const uint32_t target = (uint32_t)P0->slices;
        slice_photons = (max_photons + target - 1u) / target;
        if (slice_photons == 0u) slice_photons = 1u;
        if (slice_photons > 0x7fffffu) slice_photons = 0x7fffffu;     // (a pending entry keeps the photons left in 23 bits; at most 513 rounds then)
        rounds = (max_photons + slice_photons - 1u) / slice_photons;
        if (rounds == 0u) rounds = 1u;
    }
    // wave-uniform bookkeeping of the U unit slots: each is in a lane, in `ready`, in `pending`, empty or gone
    uint32_t sub_queue = (blockIdx.x * (uint32_t)kPoolWavesPerBlock + wave_in_group) % (uint32_t)kSubQueues;
    uint32_t used_up = 0;                       // sub-queues found used up in a row
    uint32_t n_staged = 0;                      // hit stubs waiting in the staging area
    uint32_t parked_trips = 0;                  // trips since the first of the parked lanes parked
    uint32_t n_ready = 0, ready_head = 0, n_pend = 0, n_wait = 0, n_empty = U, n_left = U;     // n_wait: pending units that wait for a predecessor; n_left: unit slots not yet retired

    // per lane: the photon it carries and the unit that photon belongs to
    // what the lane holds: one register compared against constants (three bools would live in scalar lane masks, and every
    // update under a lane-varying condition would be scalar mask arithmetic -- the scalar unit is the scarcer one here)
    // kParked: has a step length, waits for the wave's next DOM search; kParked + 1 + id: the same, and only DOM `id` is in reach
    constexpr uint32_t kVacant = 0u, kSpent = 1u, kLive = 2u, kParked = 3u;
    uint32_t st = kVacant;
    uint32_t sidx = kNoStep, ra = 0, photons_left = 0, uflags = 0;
    float parked_dist = 0.0f;                   // the step length of a parked lane (a register: 64 LDS words per wave are 2.7 ring entries)
    uint64_t rx = 0;
    Photon ph;
    ph.abs_lens_left = 0.0f;
    ph.layer = 0;

    const uint32_t wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID.wave_id
    // k_pop, k_search, k_aim, k_wait: one scalar register for the whole loop (kparams.h: k_packed)
    const uint32_t k_packed = P0->k_packed;
    CENSUS(
    if (lane == 0) atomicMin(fresh_params(P0)->census + 8, wall_clock64());
    unsigned long long c_trips = 0, c_run = 0, c_services = 0, c_creations = 0, c_created = 0, c_vacant = 0, c_polls = 0, c_parked = 0, c_searches = 0, c_chunks = 0, c_empty_ring = 0, c_hits = 0;
    // shader-clock cycles of the wave inside the service block, its publication of finished units, the unit take and the creation chunks
    unsigned long long t_service = 0, t_publish = 0, t_take = 0, t_create = 0;
    // (round 5, profiles/r05/divergence_closing.txt) one ring hand-over, measured where the kernel does it: the five 16-byte words of a
    // created photon stored (t_ring_store: the stores of one creation chunk, c_ring_stores chunks) and the hand-out block -- ballot, rank,
    // five 16-byte loads, unpacking, the wave barrier (t_hand_out, c_hand_outs blocks)
    unsigned long long t_ring_store = 0, c_ring_stores = 0, t_hand_out = 0, c_hand_outs = 0;
    const unsigned long long t_wave_start = __builtin_readcyclecounter();
    )
    // who holds what, as lane masks; taken at the end of a trip for the next one (and for the loop's exit, a plain backward branch)
    uint64_t m_spent = 0ull, m_vacant = ~0ull, m_live = 0ull;
    for (uint32_t trip = 0;; ++trip) {
        if ((trip & ((1u << kPrioShift) - 1u)) == 0u) {            // the priority changes every 2^kPrioShift trips
            // (round 6: written as two tests on the phase's bits it compiles to MORE scalar instructions -- the structuriser turns either
            // form into chains of mask moves; the switch stays)
            switch (((trip >> kPrioShift) + wave_slot) & 3u) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        CENSUS(
        ++c_trips;
        c_vacant += __popcll(m_vacant | m_spent);
        )

        // ---- service: retire the units of spent photons, create photons when the ring has room, hand out ready photons ----
        const uint32_t n_free = (uint32_t)__popcll(m_spent | m_vacant);
        // (1 <= k_pop <= 64, so this also covers "no lane holds a live photon": then all 64 are free)
        if (n_free >= (k_packed & 0xffu)) {
            const KP P = fresh_params(P0);
            WorkRecord *work = P->work;
            CENSUS(
            ++c_services;
            if (st != kLive) CENSUS_REGION(P, kCensusService);
            const unsigned long long t_s0 = __builtin_readcyclecounter();
            )
            if (m_spent != 0ull) {
                const bool mine = (st == kSpent);
                const bool finished = mine && (photons_left == 0u);
                const bool next = mine && (photons_left != 0u);
                const uint64_t m_finished = ballot(finished), m_next = ballot(next);
                if (m_finished != 0ull) {
                    CENSUS(const unsigned long long t_p0 = __builtin_readcyclecounter();)
                    // publish the finished unit (c.cl:911-912).  The last slice of a step leaves the stream's state in the
                    // converter's array for the next bunch; any other slice hands it to whoever takes the next slice:
                    // state first, then the slice counter, both write-through so that a lane on another XCD that sees
                    // the counter sees the state
                    const bool last = (uflags & kFlagLast) != 0u;
                    if (finished) {
                        if (last) P->rng_x[sidx] = rx;
                        else __hip_atomic_store(&work[sidx].x, rx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (finished && !last)
                        __hip_atomic_store(&work[sidx].done, (uflags & 0xffffu) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    n_empty += (uint32_t)__popcll(m_finished);
                    CENSUS(
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    t_publish += __builtin_readcyclecounter() - t_p0;
                    )
                }
                if (next) {     // the unit goes to `pending` with its stream where the photon left it
                    pend_store(pend, n_pend + (uint32_t)__popcll(m_next & lanes_below), sidx, rx, photons_left, uflags);
                }
                n_pend += (uint32_t)__popcll(m_next);
                if (mine) st = kVacant;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }

            if (__builtin_expect(used_up >= (uint32_t)kSubQueues, 0)) { n_left -= n_empty; n_empty = 0u; }          // the queues are dry: empty slots retire

            // photon creation: when a batch fits the ring, or when lanes would otherwise go without a photon.  A wave
            // whose pending units all wait for predecessors elsewhere looks again every fourth trip.
            const uint32_t room = R - n_ready;
            const uint32_t creatable = (n_pend - n_wait) + ((used_up < (uint32_t)kSubQueues) ? n_empty : 0u);
            const uint32_t batch = (creatable < room) ? creatable : room;
            const bool starving = (n_ready < n_free);
            const bool look_again = starving && (n_wait != 0u) && (room != 0u) && (((trip & 3u) == 0u) || (m_live == 0ull));
            if (__builtin_expect(((batch != 0u) && ((batch >= (uint32_t)P->k_new) || starving)) || look_again, 0)) {
                CENSUS(
                ++c_creations;
                const unsigned long long t_a0 = __builtin_readcyclecounter();
                )
                // (a) new units for the empty slots: one atomic per wave and round on the wave's sub-queue
                for (uint32_t round = 0; (n_empty != 0u) && (used_up < (uint32_t)kSubQueues) && (round < (uint32_t)kSubQueues + 2u); ++round) {
                    const uint32_t n_sub = (n_steps + (uint32_t)kSubQueues - 1u - sub_queue) / (uint32_t)kSubQueues;   // its steps
                    const uint32_t total_sub = n_sub * rounds;
                    const uint32_t count = (n_empty < 64u) ? n_empty : 64u;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P->queue + kQueueHeadStride * (sub_queue + 1u), count);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    const uint32_t this_queue = sub_queue;
                    if (base + count > total_sub) {                      // (also when the head has run past the end)
                        sub_queue = (sub_queue + 1u == (uint32_t)kSubQueues) ? 0u : sub_queue + 1u;
                        ++used_up;
                    } else {
                        used_up = 0;
                    }
                    bool got = false;
                    uint32_t i_new = 0, s_new = 0, left = 0, flags = 0;
                    const uint32_t unit = base + lane;
                    if ((lane < count) && (base < total_sub) && (unit < total_sub)) {
                        s_new = unit / n_sub;
                        i_new = (unit - s_new * n_sub) * (uint32_t)kSubQueues + this_queue;
                        const uint32_t num = work[i_new].step.num_photons;
                        const uint32_t first = s_new * slice_photons;
                        if (first < num) {                  // otherwise this step is used up: the slot stays empty and asks again
                            got = true;
                            const bool last = (num - first <= slice_photons);
                            left = last ? (num - first) : slice_photons;
                            flags = s_new | (last ? kFlagLast : 0u) | kFlagWaiting;
                        }
                    }
                    const uint64_t m_got = ballot(got);
                    if (got) pend_store(pend, n_pend + (uint32_t)__popcll(m_got & lanes_below), i_new, 0ull, left, flags);
                    const uint32_t n_got = (uint32_t)__popcll(m_got);
                    n_pend += n_got;
                    n_empty -= n_got;
                }
                if (used_up >= (uint32_t)kSubQueues) { n_left -= n_empty; n_empty = 0u; }      // no work is left anywhere
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

                CENSUS(
                const unsigned long long t_b0 = __builtin_readcyclecounter();
                t_take += t_b0 - t_a0;
                )
                // (b) the pending units, 64 at a time: look for the predecessor's state where needed, create the next
                // photon while the ring has room; what stays is compacted to the front of the list in its order
                uint32_t kept = 0, created = 0, still_waiting = 0;
                for (uint32_t c = 0; c < n_pend; c += 64u) {
                    CENSUS(++c_chunks;)
                    const bool have = (c + lane) < n_pend;
                    uint32_t e_sidx = 0, e_left = 0, e_flags = 0;
                    uint64_t e_rx = 0;
                    if (have) pend_load(pend, c + lane, e_sidx, e_rx, e_left, e_flags);
                    bool waiting = have && ((e_flags & kFlagWaiting) != 0u);
                    if (waiting) {
                        WorkRecord *rec = work + e_sidx;
                        const uint32_t slice = e_flags & 0xffffu;
                        const uint32_t published = (slice == 0u) ? 0u : __hip_atomic_load(&rec->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        CENSUS(++c_polls;)
                        if (published >= slice) {
                            // c.cl:458-461; slice 0 reads the state left by the previous bunch
                            e_rx = (slice == 0u) ? rec->x : __hip_atomic_load(&rec->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            e_flags &= ~kFlagWaiting;
                            waiting = false;
                        }
                    }
                    still_waiting += (uint32_t)__popcll(ballot(waiting));
                    const bool can = have && !waiting;
                    const uint64_t m_can = ballot(can);
                    const uint32_t slot = created + (uint32_t)__popcll(m_can & lanes_below);
                    const bool make = can && (slot < (R - n_ready));
                    CENSUS(unsigned long long t_q0 = 0;)
                    if (make) {
                        const WorkRecord *rec = work + e_sidx;
                        const uint32_t e_ra = rec->a;
                        const Vec3 step_dir = work_direction(&rec->step);
                        Photon born;
                        born.layer = 0;
                        create_photon<MED, TILT, FLASHER, false, FAST>(P, &rec->step, step_dir, e_rx, e_ra, born);
                        CENSUS(
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        t_q0 = __builtin_readcyclecounter();
                        )
                        uint32_t pos = ready_head + n_ready + slot;
                        if (pos >= R) pos -= R;
                        if (pos >= R) pos -= R;
                        pend_entry *q = reinterpret_cast<pend_entry *>(ready + kReadyWords * pos);      // (16-byte words)
                        q[0] = pend_entry{dm::f2u(born.px), dm::f2u(born.py), dm::f2u(born.pz), dm::f2u(born.pt)};
                        q[1] = pend_entry{dm::f2u(born.d.x), dm::f2u(born.d.y), dm::f2u(born.d.z), dm::f2u(born.inv_groupvel)};
                        q[2] = pend_entry{dm::f2u(born.abs_lens_left), dm::f2u(born.ice.sca_pow), dm::f2u(born.ice.abs_pow), dm::f2u(born.ice.abs_exp)};
                        q[3] = pend_entry{(uint32_t)born.rx_start, (uint32_t)(born.rx_start >> 32), e_sidx, e_ra};
                        q[4] = pend_entry{(uint32_t)e_rx, (uint32_t)(e_rx >> 32), e_left, e_flags | ((uint32_t)born.layer << 18)};       // (flags: 18 bits; layers < 2^14, checked by the launcher)
                        CENSUS(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");)
                    }
                    CENSUS(if (t_q0 != 0) { t_ring_store += __builtin_readcyclecounter() - t_q0; ++c_ring_stores; })
                    const bool keep = have && !make;
                    const uint64_t m_keep = ballot(keep);
                    // (every lane has read its entry above; the compacted entries land at or before the ones read)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (keep) pend_store(pend, kept + (uint32_t)__popcll(m_keep & lanes_below), e_sidx, e_rx, e_left, e_flags);
                    kept += (uint32_t)__popcll(m_keep);
                    created += (uint32_t)__popcll(ballot(make));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                n_pend = kept;
                n_wait = still_waiting;
                n_ready += created;
                CENSUS(
                c_created += created;
                t_create += __builtin_readcyclecounter() - t_b0;
                )
            }

            // ready photons for the lanes without one, oldest first
            if (n_ready != 0u) {
                CENSUS(const unsigned long long t_h0 = __builtin_readcyclecounter();)
                const bool want = (st == kVacant);
                const uint64_t m_want = ballot(want);
                const uint32_t rank = (uint32_t)__popcll(m_want & lanes_below);
                if (want && (rank < n_ready)) {
                    uint32_t pos = ready_head + rank;
                    if (pos >= R) pos -= R;
                    const pend_entry *q = reinterpret_cast<const pend_entry *>(ready + kReadyWords * pos);
                    const pend_entry q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
                    ph.px = dm::u2f(q0.x); ph.py = dm::u2f(q0.y); ph.pz = dm::u2f(q0.z); ph.pt = dm::u2f(q0.w);
                    ph.d.x = dm::u2f(q1.x); ph.d.y = dm::u2f(q1.y); ph.d.z = dm::u2f(q1.z); ph.inv_groupvel = dm::u2f(q1.w);
                    ph.abs_lens_left = dm::u2f(q2.x);
                    ph.ice.sca_pow = dm::u2f(q2.y); ph.ice.abs_pow = dm::u2f(q2.z); ph.ice.abs_exp = dm::u2f(q2.w);
                    ph.rx_start = (uint64_t)q3.x | ((uint64_t)q3.y << 32);
                    ph.layer = (int)(q4.w >> 18);
                    ph.num_scatters = 0;
                    ph.total_path = 0.0f;
                    sidx = q3.z; rx = (uint64_t)q4.x | ((uint64_t)q4.y << 32); ra = q3.w; photons_left = q4.z; uflags = q4.w & 0x3ffffu;
                    st = kLive;
                }
                uint32_t taken = (uint32_t)__popcll(m_want);
                if (taken > n_ready) taken = n_ready;
                ready_head += taken;
                if (ready_head >= R) ready_head -= R;
                n_ready -= taken;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                CENSUS(
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (taken != 0u) { t_hand_out += __builtin_readcyclecounter() - t_h0; ++c_hand_outs; }
                )
            }
            // nothing runnable in this wave: every unit it holds waits for another wave's slice
            if (ballot(st != kVacant) == 0ull) __builtin_amdgcn_s_sleep(16);
            CENSUS(t_service += __builtin_readcyclecounter() - t_s0;)
        }

        // ---- one reference loop iteration for the lanes that hold a live photon ----
        const bool run = (st == kLive);
        CENSUS(
        c_run += __popcll(ballot(run));
        c_parked += __popcll(ballot(st >= kParked));
        if (n_ready == 0u) ++c_empty_ring;
        )
        float distance = 0.0f;
        bool hit = false;
        uint32_t hit_string = 0, hit_dom = 0;
        if (run) {
            const uint32_t near_string = free_flight_bound(fresh_params(P0), ph.px, ph.py);
            distance = propagate_through_layers<MED, TILT, ANISO, FAST>(fresh_params(P0), ph, rx, ra);
            // the search cannot find a DOM closer than the nearest string cylinder: skipped when the step ends before
            // ... and of the lanes that do reach a string, most pass between two of its DOMs (second level: 3D map)
            // ... a step that can reach no other string touches this one only if it is aimed at it (segment_misses_string; not asked
            // of photons born at a DOM, which live inside the string's cylinder)
            // Asked when few lanes of the wave are at a string (a cascade in the bulk: 2 of 60); when many are (a source at a
            // string: the reference's benchmark, flashers) most of them are inside the cylinder and the question only costs.
            // (round 4, profiles/r04/aim_question.txt: two other triggers measured and dropped.  Asked whenever 1 to 8 lanes lie outside a
            // cylinder's cell (map bound > 0): the reference's benchmark.py -2.5 % -- lanes in the cylinder's cell keep the wave in the DOM map
            // whatever the others are told.  Not asked when any lane lies in a cylinder's cell: C2 -2.0 % -- a 2 m cell that touches a cylinder
            // is mostly outside it, 34 % of C2's wave trips hold such a lane, and it can very well be sent back.)
            bool at_string = !(distance < free_flight_of(near_string));
            if (!FLASHER && (uint32_t)__popcll(ballot(at_string)) <= ((k_packed >> 16) & 0xffu))
                at_string = at_string && !segment_misses_string(fresh_params(P0), ph, distance, near_string);
            if (at_string) {
                const uint32_t need = dom_search_needed<FLASHER>(fresh_params(P0), ph, distance);
                if (need != kSearchNone) {
                    st = kParked + need - kSearchFull;                   // kParked, or kParked + 1 + id
                    parked_dist = distance;
                }
            }
        }
        bool advance = (st == kLive);
        // (round 6) nobody parked -- three trips in four on cascade steps: one test instead of the count, the two thresholds and the
        // waiting counter's update (the loop is scalar-issue bound)
        const uint64_t m_parked = ballot(st >= kParked);
        if (m_parked == 0ull) {
            if (!FLASHER) parked_trips = 0u;
        } else {
            // the DOM search runs when k_search lanes are parked, or for any parked lane when nothing else can advance
            const uint32_t n_parked = (uint32_t)__popcll(m_parked);
            const uint32_t enough = (ballot(advance) == 0ull) ? 1u : ((k_packed >> 8) & 0xffu);
            // (flasher instantiations search for the first parked lane: nothing to count)
            if (!FLASHER) parked_trips = parked_trips + 1u;
            if (__builtin_expect((n_parked >= enough) || (!FLASHER && (parked_trips > (k_packed >> 24))), 0)) {
                if (!FLASHER) parked_trips = 0u;
                CENSUS(++c_searches;)
                if (KEEP && (st >= kParked)) {
                    // without STOP_PHOTONS_ON_DETECTION (c.cl:704-750): the search saves what it finds, nothing is shortened or absorbed.
                    // The lane's string mask: one word per 64 strings in the wave's LDS region.
                    const KP P = fresh_params(P0);
                    distance = parked_dist;
                    KeepSink K;
                    K.step_index = sidx;
                    K.history_n = 0u;                           // (photon histories run the classic kernel)
                    K.ring = nullptr;
                    K.string_mask = keep_mask + lane;
                    K.mask_stride = 64u;
                    K.mask_words = ((uint32_t)P->num_strings + 63u) >> 6;
                    find_collisions_keep(P, ph, distance, K);
                    st = kLive;
                    advance = true;
                }
                if (!KEEP && (st >= kParked)) {
                    distance = parked_dist;
                    // Lanes with only one DOM in reach take the search confined to it (find_collision_named: what the
                    // reference's search does for that DOM, and nothing else) -- in the flasher instantiations, and when
                    // every parked lane of the wave is of that kind: a wave that has to run the full search for one lane runs
                    // it for all of them, which costs nothing more and gives the same answer.  Measured (tools/ab_bench.py,
                    // profiles/r03/named_search_policies.txt; 1e9 photons/s for C2 / C3 / benchmark.py / C5): 0 never
                    // 3.546 / 3.053 / 2.889 / 1.915; 1 per lane, both searches in one trip 3.504 / 3.017 / 2.816 / 2.038;
                    // 2 all parked lanes or none 3.505 / 3.024 / 2.848 / 2.045; 4 = 2 in the flasher instantiations only
                    // 3.553 / 3.050 / 2.886 / 2.042.  Cascade photons that reach a string mostly arrive with steps longer than
                    // the distance to the second-nearest DOM, so their waves run the full search anyway and only pay for the
                    // second code path; photons born at a DOM live within metres of it.
                    // (the other policies: tools/experiments/named_policy.patch)
                    bool full = FLASHER ? (ballot(st == kParked) != 0ull) : true;
                    if (!full) {
                        const uint32_t id = st - (kParked + 1u);
                        const uint4 named = fresh_params(P0)->dom_named[id];
                        if (named.x != 0xffffffffu) hit = find_collision_named<FAST>(fresh_params(P0), ph, distance, id, named, hit_string, hit_dom);
                        else full = true;
                    }
                    if (full) hit = find_collision<FAST>(fresh_params(P0), ph, distance, hit_string, hit_dom);
                    st = kLive;
                    advance = true;
                }
                // ---- hit write-out (c.cl:329-385, collision c.cl:557-578) ----
                // Stubs collect in the wave's staging area ACROSS trips and leave for the photon buffer kStageRecords at
                // a time (and at the end of the kernel): one atomic on the hit counter per eight hits.  That counter is one
                // address for the whole chip and sustains about 1e8 additions per second (like the queue heads, section 5):
                // a cascade next to a string (the reference's benchmark: 4 % of the photons detected) asked for that many.
                const uint64_t hit_mask = ballot(hit);
                if (__builtin_expect(hit_mask != 0ull, 0)) {
                    const uint32_t total = (uint32_t)__popcll(hit_mask);
                    const uint32_t rank = (uint32_t)__popcll(hit_mask & lanes_below);
                    for (uint32_t done = 0; done < total;) {
                        const uint32_t space = (uint32_t)kPoolStage - n_staged;
                        const uint32_t take = (total - done < space) ? (total - done) : space;
                        if (hit && rank >= done && rank < done + take) {
                            uint32_t *st = stage + (n_staged + rank - done) * kStubWords;
                            st[0] = dm::f2u(ph.px); st[1] = dm::f2u(ph.py); st[2] = dm::f2u(ph.pz); st[3] = dm::f2u(ph.pt);
                            st[4] = dm::f2u(ph.d.x); st[5] = dm::f2u(ph.d.y); st[6] = dm::f2u(ph.d.z); st[7] = dm::f2u(distance);
                            st[8] = dm::f2u(ph.total_path); st[9] = dm::f2u(ph.abs_lens_left); st[10] = dm::f2u(ph.inv_groupvel);
                            st[11] = ph.num_scatters; st[12] = sidx;
                            st[13] = (uint32_t)ph.rx_start; st[14] = (uint32_t)(ph.rx_start >> 32);
                            st[15] = (hit_string & 0xffffu) | (hit_dom << 16);
                        }
                        n_staged += take;
                        done += take;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (n_staged == (uint32_t)kPoolStage) {
                            flush_hit_stubs(fresh_params(P0), stage, n_staged, lane);
                            n_staged = 0u;
                        }
                    }
                }
            }
        }
        if (advance) {
            if (hit) ph.abs_lens_left = 0.0f;                                   // c.cl:741-744
            ph.px += ph.d.x * distance;
            ph.py += ph.d.y * distance;
            ph.pz += ph.d.z * distance;
            ph.pt += ph.inv_groupvel * distance;
            ph.total_path += distance;
            if (ph.abs_lens_left < kEpsilon) {
                --photons_left;                                                 // absorbed or detected
                st = kSpent;
            } else {
                const KP P = fresh_params(P0);
                if (ANISO && P->has_pre) apply_matrix(P->pre, P->pre_renorm, ph.d, FAST || (P->div_ok & kFastMatrices) != 0u);
                const float cos_s = scattering_cos<FAST>(P, rx, ra);
                const float sin_s = dm::sqrt_near_(1.0f - sqr(cos_s));       // |cos_s| <= 1: 0 or >= 2^-24
                scatter_direction(cos_s, sin_s, ph.d, rng_co(rx, ra));
                if (ANISO && P->has_post) apply_matrix(P->post, P->post_renorm, ph.d, FAST || (P->div_ok & kFastMatrices) != 0u);
                ++ph.num_scatters;
            }
        }
        m_spent = ballot(st == kSpent);
        m_vacant = ballot(st == kVacant);
        m_live = ballot(st >= kLive);
        if (__builtin_expect((m_live | m_spent | (uint64_t)n_left) == 0ull, 0)) break;                // every unit slot has been retired (one test: the loop is scalar-issue bound)
    }
    if (n_staged != 0u) flush_hit_stubs(fresh_params(P0), stage, n_staged, lane);
    CENSUS(
    if (lane == 0) {
        unsigned long long *d = fresh_params(P0)->census;
        atomicAdd(d + 0, c_trips); atomicAdd(d + 1, c_run); atomicAdd(d + 2, c_services); atomicAdd(d + 3, c_creations);
        atomicAdd(d + 4, c_created); atomicAdd(d + 5, c_vacant); atomicAdd(d + 6, c_polls); atomicAdd(d + 7, c_parked);
        atomicAdd(d + 9, c_searches); atomicAdd(d + 10, c_chunks); atomicAdd(d + 11, c_empty_ring);
        atomicAdd(d + 12, t_service); atomicAdd(d + 13, t_publish); atomicAdd(d + 14, t_take); atomicAdd(d + 15, t_create);
        atomicAdd(d + 24600, (unsigned long long)__builtin_readcyclecounter() - t_wave_start);
        atomicAdd(d + 24601, t_ring_store); atomicAdd(d + 24602, c_ring_stores); atomicAdd(d + 24603, t_hand_out); atomicAdd(d + 24604, c_hand_outs);
        const uint32_t w = blockIdx.x * (uint32_t)kPoolWavesPerBlock + wave_in_group;
        d[16 + 3 * w] = wall_clock64();
        d[16 + 3 * w + 1] = 0;
        d[16 + 3 * w + 2] = c_trips;
    }
    )
}

// ring entries per wave that fit beside a table image of `table_words` words (two workgroups per CU share 160 KB; the image is per
// workgroup, the rest goes to the waves' pools); keep_strings: the detector's strings without STOP_PHOTONS_ON_DETECTION, else 0
static int pool_ring_that_fits(uint32_t table_words, uint32_t keep_strings)
{
    const int budget_words = (160 * 1024 / CLSIMHIP_POOL_GROUPS - 4096 / CLSIMHIP_POOL_GROUPS) / 4 - (int)((table_words + 3u) & ~3u);         // per workgroup
    const int per_wave = (budget_words / kPoolWavesPerBlock) & ~3;                      // (a wave's region is a multiple of 16 bytes)
    return (per_wave - (int)kPoolFixedWords - (int)pool_keep_extra_words(keep_strings) - (int)kPendWords * 64) / (int)(kReadyWords + kPendWords);
}

// ---- host-side launcher (called from launch_prop_kernel) ----
hipError_t launch_scan_steps(const KParams &P, hipStream_t stream);
hipError_t launch_assemble_hits(const KParams &P, bool flasher, int device, hipStream_t stream);

template <int MED, bool TILT, bool ANISO, bool FLASHER, bool FAST, bool KEEP>
static hipError_t launch_pool_variant(const KParams &Pin, hipStream_t stream, int grid_wanted = 0)
{
    KParams P = Pin;
    int dev = 0;
    {
        const hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
    }
    // LDS: two workgroups per CU share 160 KB; the image is per workgroup, the rest goes to the waves' pools
    int R = P.pool_ready;
    {
        const int fit = pool_ring_that_fits(P.table_words, KEEP ? (uint32_t)P.num_strings : 0u);
        if (R <= 0 || R > fit) R = fit;
        if (R > 64) R = 64;
        // a ring the caller asked for ("pool_ring") below the smallest one the kernel runs with is raised to it; an image
        // that leaves no room even for that never gets here: pool_kernel_fits() (below, the same arithmetic with the
        // larger threshold from which the pooled kernel pays) sends its bunches to the classic kernel
        if (R < kPoolMinReady) R = kPoolMinReady;
        if (R > fit) return hipErrorInvalidValue;
        P.pool_ready = R;
    }
    const size_t lds_bytes = (size_t)(((P.table_words + 3u) & ~3u) + kPoolWavesPerBlock * pool_wave_words((uint32_t)R, KEEP ? pool_keep_extra_words((uint32_t)P.num_strings) : 0u)) * 4;
    struct Plan { int cus = 0, resident = 0; };
    static std::mutex plan_mutex;
    static std::map<std::pair<int, size_t>, Plan> plans;
    Plan plan;
    {
        std::lock_guard<std::mutex> lk(plan_mutex);
        Plan &pl = plans[std::make_pair(dev, lds_bytes)];        // (per instantiation: the map is a static of this template)
        if (pl.resident == 0) {
            int cus = 0, per_cu = 0;
            hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (e == hipSuccess && lds_bytes > 64 * 1024)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(&prop_pool_kernel<MED, TILT, ANISO, FLASHER, FAST, KEEP>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e == hipSuccess)
                e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, prop_pool_kernel<MED, TILT, ANISO, FLASHER, FAST, KEEP>, kPoolBlock, lds_bytes);
            if (e != hipSuccess) return e;
            if (per_cu < 1) per_cu = 1;
            if (cus < 1) cus = 1;
            pl.cus = cus;
            pl.resident = cus * per_cu;
        }
        plan = pl;
    }
    uint32_t grid = (uint32_t)plan.resident;
    if (P.chip_share > 1) grid = (grid / (uint32_t)P.chip_share > 0u) ? grid / (uint32_t)P.chip_share : 1u;      // concurrent launches share the chip
    if (grid_wanted >= 1 && grid_wanted <= plan.resident) grid = (uint32_t)grid_wanted;      // clsimhip_set_tuning("grid")
    // Fewer steps than the grid has unit slots: smaller rings on every CU rather than full rings on fewer CUs (round 4: with the ring of 45 a
    // bunch of 625 000 flasher steps filled 478 of the 512 workgroups; a ring entry is worth 0.28 %, a workgroup 0.2 %) -- unless the ring
    // was asked for ("pool_ring") or would fall below the size from which the pooled kernel pays
    size_t lds_launch = lds_bytes;
    if (Pin.pool_ready <= 0 && (uint64_t)grid * kPoolWavesPerBlock * (64u + (uint32_t)R) > (uint64_t)P.n_steps) {
        const int smaller = (int)((uint64_t)P.n_steps / ((uint64_t)grid * kPoolWavesPerBlock)) - 64;
        // (only a little smaller: cascade steps, photons/s, smaller rings on all CUs / full rings on fewer -- 0.49M steps, ring 16: 2.62 / 2.90e9; 0.56M, ring 26:
        // 3.12 / 3.18; 0.62M, ring 37: 3.42 / 3.42; 625 000 flasher steps, ring 37: 2.30 / 2.23)
        if (smaller >= kPoolWorthwhileReady && smaller < R && 5 * smaller >= 4 * R) {
            R = smaller;
            P.pool_ready = R;
            lds_launch = (size_t)(((P.table_words + 3u) & ~3u) + kPoolWavesPerBlock * pool_wave_words((uint32_t)R, KEEP ? pool_keep_extra_words((uint32_t)P.num_strings) : 0u)) * 4;
        }
    }
    // never more unit slots than steps
    const uint32_t slots_per_group = (uint32_t)kPoolWavesPerBlock * (64u + (uint32_t)R);
    const uint32_t needed = (P.n_steps + slots_per_group - 1u) / slots_per_group;
    if (needed < grid) grid = needed;
    {
        const double r = (double)P.n_steps / ((double)grid * slots_per_group);
        // (1M steps: 8 slices 2.76e9 photons/s, 12: 2.84, 16: 2.84, 24: 2.85, 32: 2.87; fabric traffic 18.6 / 19.2 / 20.7 GB per
        // launch at 12 / 16 / 32 slices: the last per cent of speed is not worth a tenth more traffic)
        // (r < 1 only because the grid was cut to the workgroups the bunch fills: then r > 1 - 1 / grid and the bunch is sliced like a larger
        // one -- with whole steps a bunch just below the chip's unit slots ran 10 % slower than one just above, profiles/r04/ab_ring_size.txt)
        if (P.slices <= 0) P.slices = (r < 0.95) ? 1 : 16;
        // lanes parked per DOM search: with the two-level proximity filter about 1 % of the lanes need one per trip (cascade
        // steps: 3 parked lanes 2.55e9 photons/s, 1: 2.49, 5: 2.53, 8: 2.27 at 1M steps); photons born at a DOM need one on
        // most trips whatever the filter (flasher steps: 3 parked lanes 1.50e9, 5: 1.58, 7: 1.615, 9: 1.625, 12: 1.616, 16: 1.57)
        // Since round 3 the filter itself discards the photons that are still inside the DOM they were born in (dom_search_needed<INSIDE>):
        // flasher steps now need 0.005 searches per trip instead of 0.28, and a lane that waits for company waits long
        // (2.6M flasher steps: 1 parked lane 2.32e9 photons/s, 2: 2.30, 3: 2.28, 4: 2.25, 8: 2.13; profiles/r03/c5_inside_filter.txt)
        // Cascade steps, since the filter asks whether a photon that passes a string is aimed at it (segment_misses_string): 0.009
        // searches per trip in the bulk (was 0.076), where a parked lane would wait a hundred trips for two more -- so it waits
        // k_wait trips at most; next to a source on a string (the reference's benchmark.py in this detector) lanes arrive every
        // other trip and a batch of 5 fills in time.  1M cascade steps / benchmark.py, 1e9 photons/s: k_search, k_wait = 3, 4: 3.72 / 2.81;
        // 3, 16: 3.71 / 2.84; 5, 16: 3.72 / 2.87; 8, 16: 3.71 / 2.85; 1, -: 3.72 / 2.68; 3, none: 3.39 / 2.84
        // (profiles/r03/string_aimed_filter.txt)
        if (P.k_search <= 0) P.k_search = (r < 0.95) ? 1 : (FLASHER ? 1 : 5);
        // (round 2, flasher steps, 2.6M: 4 free lanes per service 1.58e9 photons/s, 6: 1.60, 8: 1.61; cascade steps: 3 and 4 3.02e9, 6: 3.00, 8: 2.96.
        // Round 4, ring of 45 and the inside-a-DOM filter: flasher steps 3, 4, 5: 2.48e9, 6: 2.47, 8: 2.44, 10: 2.41, 12: 2.37; cascade steps
        // 3: 3.99, 4 - 6: 4.01 - 4.02, 8: 3.99; SPICE-Lea 4, 5: 3.42, 6: 3.41; benchmark.py 4: 3.00, 5: 2.99, 6: 2.98 -- profiles/r04/scan_k_pop.txt)
        if (P.k_wait < 0) P.k_wait = 16;          // (0 is honoured: search as soon as a lane is parked)
        if (P.k_aim < 0) P.k_aim = 8;             // (0 is honoured: the string-aimed level is off)
        if (P.k_pop <= 0) P.k_pop = 4;
        if (P.k_pop > 64) P.k_pop = 64;
        // create when the ring is down to its last entry: the batches are what makes creation cheap per photon
        // (ring of 34: threshold 20 2.76e9 photons/s, 26: 2.81, 30: 2.84, 33: 2.85)
        if (P.k_new <= 0 || P.k_new > R) P.k_new = (R > 8) ? R - 1 : R;
        if (P.k_wait > 255) P.k_wait = 255;
        if (P.k_search > 64) P.k_search = 64;
        if (P.k_aim > 64) P.k_aim = 64;
        P.k_packed = (uint32_t)P.k_pop | ((uint32_t)P.k_search << 8) | ((uint32_t)P.k_aim << 16) | ((uint32_t)P.k_wait << 24);
        if (P.slices > 0xffff) P.slices = 0xffff;
        if ((uint64_t)P.n_steps * (uint64_t)P.slices >= 0x7fffffffull) P.slices = 1;    // 32-bit unit counters
    }
    hipError_t err = launch_scan_steps(P, stream);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL((prop_pool_kernel<MED, TILT, ANISO, FLASHER, FAST, KEEP>), dim3(grid), dim3(kPoolBlock), lds_launch, stream, P);
    err = hipGetLastError();
    if (err != hipSuccess) return err;
    return launch_assemble_hits(P, FLASHER, dev, stream);
}

#ifndef CLSIMHIP_POOL_KEEP_UNIT
#define CLSIMHIP_POOL_LAUNCHER launch_pool_kernel
#define CLSIMHIP_POOL_KEEP false
#else       // prop_pool_keep_kernel.hip: the instantiations without STOP_PHOTONS_ON_DETECTION
#define CLSIMHIP_POOL_LAUNCHER launch_pool_keep_kernel
#define CLSIMHIP_POOL_KEEP true
#endif
hipError_t CLSIMHIP_POOL_LAUNCHER(const KParams &P, const KVariant &v, hipStream_t stream)
{
    if (P.n_steps == 0) return hipSuccess;
    if (v.lengths < CLSIMHIP_LENGTHS_CONSTANT || v.lengths > CLSIMHIP_LENGTHS_TABLE) return hipErrorInvalidValue;
    if (v.lengths == CLSIMHIP_LENGTHS_TABLE && (!P.len_table || P.len_tab_n < 2)) return hipErrorInvalidValue;
    if (P.history_n != 0 || v.tabulate || (v.keep_detected != CLSIMHIP_POOL_KEEP)) return hipErrorInvalidValue;
    if (P.num_layers >= (1 << 14)) return hipErrorInvalidValue;          // (a ring entry keeps the carried layer index in 14 bits: pool_kernel_fits() says so first)
    if (P.n_steps > kPoolIndexMask) return hipErrorInvalidValue;          // (a pending entry keeps the step index in 23 bits: Converter::pooled_for() says so first)
    const int key = 8 * v.lengths + (v.tilt ? 4 : 0) + (v.aniso ? 2 : 0) + (v.flasher ? 1 : 0);
    // clsimhip_set_tuning("generic_kernels", 1): the generic instantiation also where Compile() found every proof (tests compare the two)
    const bool fast = v.fast && !v.generic_only;
    switch (key) {
#define CASE(k, a, b, c, d) case k: return fast ? launch_pool_variant<a, b, c, d, true, CLSIMHIP_POOL_KEEP>(P, stream, v.grid) : launch_pool_variant<a, b, c, d, false, CLSIMHIP_POOL_KEEP>(P, stream, v.grid);
#define CASES(m) \
    CASE(8 * m + 0, m, false, false, false) CASE(8 * m + 1, m, false, false, true) \
    CASE(8 * m + 2, m, false, true, false)  CASE(8 * m + 3, m, false, true, true)  \
    CASE(8 * m + 4, m, true, false, false)  CASE(8 * m + 5, m, true, false, true)  \
    CASE(8 * m + 6, m, true, true, false)   CASE(8 * m + 7, m, true, true, true)
    CASES(CLSIMHIP_LENGTHS_CONSTANT) CASES(CLSIMHIP_LENGTHS_ICECUBE) CASES(CLSIMHIP_LENGTHS_TABLE)
#undef CASES
#undef CASE
    }
    return hipErrorInvalidValue;
}

#ifndef CLSIMHIP_POOL_KEEP_UNIT
// the largest bunch the pooled kernel's 23-bit step index can address (clsimhip_set_tuning("pool_max_steps") lowers the GUARD for tests:
// larger bunches then take the classic kernel, exactly what a bunch beyond 2^23 - 1 does)
size_t pool_kernel_max_steps() { return (size_t{1} << kPoolIndexBits) - 1; }
// does the pooled kernel pay for this table image (its waves need at least kPoolWorthwhileReady ring entries)?  keep_strings: the
// number of strings when the converter runs without STOP_PHOTONS_ON_DETECTION (the search's string masks share the pool's LDS), else 0
bool pool_kernel_fits(uint32_t table_words, uint32_t keep_strings, int num_layers)
{
    if (num_layers >= (1 << 14)) return false;
    static_assert(kPoolWorthwhileReady >= kPoolMinReady, "an image the pooled kernel is chosen for must be one it can run");
    return pool_ring_that_fits(table_words, keep_strings) >= kPoolWorthwhileReady;
}
#endif

} // namespace clsimhip
