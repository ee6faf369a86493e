// ls_tuning.h -- tuning / ablation knobs of the experiment scripts under tools/ (sweep_*.sh, exp_*.sh, upload_sweep.py).
// The shipped library never reads the environment for them: tune_int() is the built-in default unless the library was
// built with -DLS_EXPERIMENTAL (make -C lidarshooter_amd/csrc EXPERIMENTAL=1), which is what those scripts do.
#pragma once

#include <cstdlib>

namespace lsi {

inline int tune_int(const char *name, int def)
{
#ifdef LS_EXPERIMENTAL
    if (const char *e = std::getenv(name)) return std::atoi(e);
#else
    (void)name;
#endif
    return def;
}

}  // namespace lsi
