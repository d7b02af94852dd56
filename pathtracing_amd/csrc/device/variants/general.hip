// The general kernels: every feature compiled in, scene tables read from memory (scenes too large for the LDS copies).
#include "define.h"

namespace ptk {
PTK_BOUNCE_VARIANT(true, true, false, true, false)
PTK_BOUNCE_VARIANT(true, true, true, true, false)
PTK_TAIL_VARIANT(true, true, true, false)
PTK_RESUME_VARIANT(true, true, false)
} // namespace ptk
