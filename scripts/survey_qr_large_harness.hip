// experiment: instantiate ONE qr-family kernel, contiguous-only kind (compile time / resources)
#define NFM_QR_PART 99
#include "nfm_qr.hip"
using namespace nfm;
#ifndef EN
#define EN 16
#endif
#ifndef ET
#define ET float
#endif
#ifndef EOP
#define EOP EigSymOp<ET, EN, false, false>
#endif
extern "C" int exp_op(int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *b, const nfm_operand *o, void *stream)
{
    QrParams p = mkparams(EN, 1, 1, 0, 1024, 1e-32);
    return rec_launch<ET, EOP, true>(a, b, nullptr, o, no, ni, p, stream);
}
