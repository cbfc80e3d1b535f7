"""CPU-suite lint: every extern "C" entry point of csrc/*.hip that takes a handle must select the
handle's device (hipSetDevice) BEFORE its first HIP runtime call -- allocation, attribute, launch, event, stream, copy --
or its first call of a file-local helper that makes one.  On a one-GPU box a forgotten hipSetDevice cannot show (the current
device is always 0); on an 8-GPU node the call would land on whatever device the calling thread used last."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "csrc")

HANDLE_TYPES = ("orb_extractor", "orb_matcher", "orb_vocab", "orb_multi", "orb_multi_db")
# HIP calls that are legitimate before a device is selected (they take or return the device explicitly, or touch no device)
NEUTRAL = {"hipSetDevice", "hipGetDeviceCount", "hipGetLastError", "hipGetErrorString", "hipDeviceGetAttribute", "hipHostFree",
           "hipSuccess", "hipError_t", "hipStream_t", "hipEvent_t", "hipErrorNotReady"}
HIP_CALL = re.compile(r"\b(hip[A-Z]\w*)\s*\(|(<<<)")


def _strip(src):
    src = src.replace('extern "C"', "EXTERN_C")
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r'"(?:\\.|[^"\\])*"', '""', src)


def _body(src, open_brace):
    depth, i = 0, open_brace
    while i < len(src):
        if src[i] == "{":
            depth += 1
        elif src[i] == "}":
            depth -= 1
            if depth == 0:
                return src[open_brace:i + 1]
        i += 1
    raise AssertionError("unbalanced braces")


DEF = re.compile(r"(?m)^(?P<kind>EXTERN_C|static|inline)?\s*(?P<ret>(?:const\s+)?[A-Za-z_][\w:<>]*[\s\*&]+)(?P<name>[A-Za-z_]\w*)\s*\((?P<params>[^;{}()]*(?:\([^()]*\)[^;{}()]*)*)\)\s*\{")


def _functions(src):
    """(is_extern_c, name, params, body) of the function definitions at file scope (kernels and templates are not matched:
    device code makes no runtime calls)"""
    out = []
    for m in DEF.finditer(src):
        if m.group("ret").split()[0] in ("return", "else", "new", "struct", "class", "namespace", "typedef", "template", "__global__"):
            continue
        out.append((m.group("kind") == "EXTERN_C", m.group("name"), m.group("params"), _body(src, m.end() - 1)))
    return out


def _first_device_call(body, helpers):
    """the first HIP runtime call / kernel launch / hip-calling helper in the body, in source order"""
    best = None
    for m in HIP_CALL.finditer(body):
        name = m.group(1) or "<<<"
        if name in NEUTRAL and name != "hipSetDevice":
            continue
        best = (m.start(), name)
        break
    for h in helpers:
        m = re.search(r"\b%s\s*\(" % re.escape(h), body)
        if m and (best is None or m.start() < best[0]):
            best = (m.start(), h)
    return best


def test_entry_points_select_the_device_first():
    problems, checked = [], 0
    files = {path: _functions(_strip(open(path).read())) for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")))}
    # helpers (not entry points) that make HIP calls themselves, directly or through another helper -- in any file: the launch
    # wrappers of one translation unit are called from another
    helpers = set()
    changed = True
    while changed:
        changed = False
        for funcs in files.values():
            for ext, name, params, body in funcs:
                if ext or name in helpers:
                    continue
                if _first_device_call(body[1:], helpers - {name}) is not None:
                    helpers.add(name)
                    changed = True
    assert "orb_launch_quadtree" in helpers and "build_geometry" in helpers, sorted(helpers)
    # a helper whose own first device call is hipSetDevice selects its device itself (match_host, build_shard, ...): calling it
    # first is fine
    bodies = {name: body for funcs in files.values() for ext, name, params, body in funcs if not ext}
    selfsel = {h for h in helpers if (_first_device_call(bodies[h][1:], helpers - {h}) or (0, ""))[1] == "hipSetDevice"}
    helpers -= selfsel
    for path, funcs in files.items():
        for ext, name, params, body in funcs:
            if not ext:
                continue
            takes_handle = any(re.search(r"\b%s\s*\*" % t, params) for t in HANDLE_TYPES)
            if not takes_handle or name.endswith("_destroy"):
                continue                                            # (destroy: checked by hand -- sets the device first, tolerates a dead one)
            first = _first_device_call(body, helpers)
            checked += 1
            if first is not None and first[1] != "hipSetDevice":
                problems.append("%s: %s() reaches %s before hipSetDevice" % (os.path.basename(path), name, first[1]))
    assert checked > 40, checked                                    # the scan did find the entry points
    assert not problems, "\n".join(problems)


def test_the_lint_sees_a_planted_mistake():
    src = _strip('''
static int helper(orb_matcher* m) { return hipMalloc(&m->p, 4); }
extern "C" int orb_bad(orb_matcher* m, int n)
{
    if (!m) return -1;
    int rc = helper(m);
    ORB_HIP_TRY(hipSetDevice(m->device));
    return rc;
}
extern "C" int orb_good(orb_matcher* m)
{
    ORB_HIP_TRY(hipSetDevice(m->device));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, m->stream);
    return helper(m);
}''')
    funcs = _functions(src)
    assert [f[1] for f in funcs] == ["helper", "orb_bad", "orb_good"] and [f[0] for f in funcs] == [False, True, True]
    assert _first_device_call(funcs[1][3], {"helper"})[1] == "helper"
    assert _first_device_call(funcs[2][3], {"helper"})[1] == "hipSetDevice"


def test_no_legacy_stream_operations_in_the_library():
    """hipMemcpy / hipMemset / hipMemcpy2D (the synchronous forms) and hipDeviceSynchronize go through the legacy null stream.
    On ROCm 7.2 they FAIL while any stream of the process is being captured into a graph -- and invalidate that capture --,
    even a thread-local capture of another thread's non-blocking stream: with two extractor handles on two threads
    (reference src/Frame.cc:82-85) one thread's table upload met the other thread's capture of its single-frame chain.
    The library copies through a stream of the caller's (orb_copy_blocking / orb_fill_blocking, csrc/orb_common.h)."""
    import glob
    import re
    bad = []
    pat = re.compile(r"\b(hipMemcpy|hipMemset|hipMemcpy2D|hipMemcpyToSymbol|hipMemcpyFromSymbol|hipDeviceSynchronize)\s*\(")
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        for ln, line in enumerate(_strip(open(path).read()).splitlines(), 1):
            if pat.search(line):
                bad.append("%s:%d: %s" % (os.path.basename(path), ln, line.strip()[:100]))
    assert not bad, "\n".join(bad)
