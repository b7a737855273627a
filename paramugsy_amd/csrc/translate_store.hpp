// translate_store.hpp -- device-resident row tables shared by the translate job and untranslate.
#pragma once

#include "pm_internal.hpp"
#include "translate_device.hpp"

namespace pm {

struct RowsStore {
  DevBuf range, length, gap_off, gaps, pre, bad, raw_s, raw_e, raw_gs, raw_ge;
  DevBuf range32, length32, gaps32, pre32; // the same tables as int (only with upload_rows(..., maxabs != null))
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
  // positions in 64 bits, everything else from the int tables (translate_device.hpp: P = long long, I = int)
  RowsT<int, i64> view_mixed() const {
    RowsT<int, i64> d;
    d.n = n;
    d.range = (const R2 *)range.p;
    d.length = (const int *)length32.p;
    d.gap_off = (const i64 *)gap_off.p;
    d.gaps = (const R2T<int> *)gaps32.p;
    d.pre = (const int *)pre32.p;
    d.bad = (const int *)bad.p;
    return d;
  }
  RowsT<int> view32() const {
    RowsT<int> d;
    d.n = n;
    d.range = (const R2T<int> *)range32.p;
    d.length = (const int *)length32.p;
    d.gap_off = (const i64 *)gap_off.p;
    d.gaps = (const R2T<int> *)gaps32.p;
    d.pre = (const int *)pre32.p;
    d.bad = (const int *)bad.p;
    return d;
  }
};

// Upload one side's rows and run prepare_rows_kernel (interleaved gaps, prefix table, validation).  With `maxabs` (a
// device word, zeroed by the caller) the int copies of the tables are built as well and the OR of every magnitude in the
// tables is accumulated into maxabs[0] (sequence positions: the rows' starts and ends) and maxabs[1] (everything in columns: lengths,
// gaps, prefix sums, and the rows' spans |end - start| + 1).
int upload_rows(const pm_rows_t *h, RowsStore &s, hipStream_t stream, unsigned long long *maxabs = nullptr);

} // namespace pm
