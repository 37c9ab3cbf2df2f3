"""CPU checks that the PRODUCT library is sealed: no environment switches outside -DBOCF_PROBES, no kernels kept
"for A/B", every collective reachable by every rank."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bocf_amd", "csrc")


def _product_text(path):
    """The text of a source file with every #ifdef BOCF_PROBES ... (#else) ... #endif region reduced to its #else part."""
    out, stack = [], []          # stack entries: [is_probes_block, in_else]
    for line in open(path):
        t = line.strip()
        if re.match(r"#\s*if", t):
            stack.append([bool(re.match(r"#\s*ifdef\s+BOCF_PROBES\b", t)), False])
            continue
        if re.match(r"#\s*else", t) and stack:
            stack[-1][1] = True
            continue
        if re.match(r"#\s*endif", t) and stack:
            stack.pop()
            continue
        if any(p and not e for p, e in stack):
            continue
        out.append(line)
    return "".join(out)


def test_no_getenv_in_the_product_library():
    offenders = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            text = _product_text(os.path.join(CSRC, f))
            text = re.sub(r"//[^\n]*", "", text)
            if "getenv" in text:
                offenders.append(f)
    assert not offenders, offenders


def test_dead_kernels_are_gone():
    allsrc = "".join(open(os.path.join(CSRC, f)).read() for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for name in ("potrf_diag_kernel", "potrf_diag_mfma_kernel", "chol128_regs", "inv128_regs", "set_gemm_store_waves", "BOCF_INFER_SCALAR",
                 "BOCF_KBUILD_SCALAR", "BOCF_ACQ_GENERIC", "BOCF_SMALL_SCALAR"):
        assert not re.search(r"\b%s\s*\(" % name, allsrc) and ('"%s"' % name) not in allsrc, name
    # the two-buffer 256-row GEMM and the slower tile orders live in the probes build only
    gemm = _product_text(os.path.join(CSRC, "gemm_f64.hip"))
    assert "gemm_tn_f64_sumsq256_kernel" not in gemm and "g.swizzle >= 100" not in gemm


def test_collectives_are_reached_by_every_rank():
    """bocf_global_topk: nothing may return between the argument checks (identical on every rank) and the all-reduce except the rank
    that aborts the communicator; the local status travels in the packed buffer's spare slot."""
    text = open(os.path.join(CSRC, "comm.hip")).read()
    body = text[text.index('extern "C" int bocf_global_topk('):]
    body = body[:body.index("g_rccl.AllReduce(")]
    after_args = body[body.index("const int world"):]
    returns = [l.strip() for l in after_args.split("\n") if re.search(r"\breturn\b", l)]
    assert returns == ["return -1;"], returns            # the one behind bocf_comm_abort
    assert "bocf_comm_abort" in after_args
