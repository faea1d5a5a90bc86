// sbg_core.hip -- library-wide state of libsbg_hip.so: version and the thread-local error slot.
#include "sbg_common.h"

std::string& sbg_err_slot()
{
    static thread_local std::string slot;
    return slot;
}

extern "C" int sbg_version(void) { return 1; }

extern "C" const char* sbg_last_error(void) { return sbg_err_slot().c_str(); }
