"""Error model of the C ABI (reference src/errors.rs:3-50, src/utils.rs:52-118)."""
import ctypes as C

from ._lib import lib

CODES = {
    0: "NoError", 1: "Panic", 2: "Internal", 3: "Msg", 4: "Unknown",
    101: "MismatchKSizes", 102: "MismatchDNAProt", 103: "MismatchMaxHash", 104: "MismatchSeed",
    1101: "InvalidDNA", 1102: "InvalidProt",
    100001: "Io", 100002: "Utf8Error", 100003: "ParseInt", 100004: "SerdeError",
}


class SourmashError(Exception):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (CODES.get(code, "?"), code, message))
        self.code = code
        self.message = message


def take_str(s):
    """Copy a SourmashStr to bytes and free it."""
    out = C.string_at(s.data, s.len) if s.data and s.len else b""
    lib().sourmash_str_free(C.byref(s))
    return out


def check():
    """Raise the thread's pending error, clearing the slot (how Python sourmash's rustcall works)."""
    L = lib()
    code = L.sourmash_err_get_last_code()
    if code:
        msg = take_str(L.sourmash_err_get_last_message()).decode("utf-8", "replace")
        L.sourmash_err_clear()
        raise SourmashError(code, msg)


def call(fn, *args):
    L = lib()
    L.sourmash_err_clear()
    r = fn(*args)
    check()
    return r
