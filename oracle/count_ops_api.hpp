/*
 * count_ops_api.hpp -- TEST / MEASUREMENT INFRASTRUCTURE (counting build of the oracle, see count_ops.hpp): the totals,
 * included at the end of clsim_oracle.c.
 */
static oc_counters oc_total;
thread_local oc_counters oc_tl;
thread_local int oc_region_now = OCR_OTHER;

void oc_fold_thread_counters(void)       /* callers serialise (an omp critical section, or a serial driver) */
{
    for (int r = 0; r < OCR_NUM_REGIONS; ++r)
        for (int o = 0; o < OC_NUM_OPS; ++o) { oc_total.ops[r][o] += oc_tl.ops[r][o]; oc_tl.ops[r][o] = 0; }
    for (int e = 0; e < OCE_NUM_EVENTS; ++e) { oc_total.events[e] += oc_tl.events[e]; oc_tl.events[e] = 0; }
}
extern "C" void oracle_count_reset(void) { memset(&oc_total, 0, sizeof oc_total); memset(&oc_tl, 0, sizeof oc_tl); }
/* ops: OCR_NUM_REGIONS x OC_NUM_OPS, events: OCE_NUM_EVENTS (row-major, the enumerations of count_ops.hpp) */
extern "C" void oracle_count_get(uint64_t *ops, uint64_t *events)
{
    memcpy(ops, oc_total.ops, sizeof oc_total.ops);
    memcpy(events, oc_total.events, sizeof oc_total.events);
}
extern "C" void oracle_count_shape(int32_t *regions, int32_t *ops, int32_t *events) { *regions = OCR_NUM_REGIONS; *ops = OC_NUM_OPS; *events = OCE_NUM_EVENTS; }
