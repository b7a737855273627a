// translate_store.hpp -- device-resident row tables shared by the translate job and untranslate.
#pragma once

#include "pm_internal.hpp"
#include "translate_device.hpp"

namespace pm {

struct RowsStore {
  DevBuf range, length, gap_off, gaps, pre, bad, raw_s, raw_e, raw_gs, raw_ge;
  i64 n = 0, G = 0;
  RowsD view() const {
    RowsD d;
    d.n = n;
    d.range = (const R2 *)range.p;
    d.length = (const i64 *)length.p;
    d.gap_off = (const i64 *)gap_off.p;
    d.gaps = (const R2 *)gaps.p;
    d.pre = (const i64 *)pre.p;
    d.bad = (const int *)bad.p;
    return d;
  }
};

// Upload one side's rows and run prepare_rows_kernel (interleaved gaps, prefix table, validation).
int upload_rows(const pm_rows_t *h, RowsStore &s, hipStream_t stream);

} // namespace pm
