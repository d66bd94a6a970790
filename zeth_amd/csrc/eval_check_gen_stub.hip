// Placeholder registry used until tools/gen_eval_check.py has produced eval_check_gen.hip.
#include "circuit.h"
namespace zkh {
#ifndef ZKH_HAVE_GENERATED_EVAL_CHECK
const CompiledEvalCheck* find_compiled_eval_check(uint64_t) { return nullptr; }
#endif
}
