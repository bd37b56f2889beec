// xbit_launch_all.h — a call that goes out as SEVERAL launches of one kernel family (rows in chunks of 16 or 32) must not
// discover at its last chunk that the family does not take it: the chunks already launched would then be computed a second
// time by the fallback.  launch_all_or_nothing runs the body once as a dry run (the launch rules run, nothing is issued:
// xbit_host.h, ProfileState::dry_run) and for real only when every launch of it was taken.
#pragma once

#include "xbit_host.h"

namespace xbit {

template <class F>
int launch_all_or_nothing(F&& body) {
  ProfileState& ps = profile_state();
  if (ps.dry_run) return body();  // the caller is recording already: once is enough
  const size_t mark = ps.records.size();
  ps.dry_run = true;
  const int st = body();
  ps.dry_run = false;
  ps.records.resize(mark);
  if (st != XBIT_OK) return st;  // not eligible somewhere (or a bad argument): nothing has been launched
  return body();
}

}  // namespace xbit
