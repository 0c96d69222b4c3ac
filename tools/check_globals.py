"""Undefined-global check (no pyflakes in the image): every LOAD_GLOBAL of every function of the given modules must be a
module global or a builtin.  python tools/check_globals.py benchlib.common benchlib.apply ..."""
import builtins
import dis
import importlib
import sys
import types


def walk(code):
    yield code
    for c in code.co_consts:
        if isinstance(c, types.CodeType):
            yield from walk(c)


bad = 0
sys.path.insert(0, ".")
for name in sys.argv[1:]:
    m = importlib.import_module(name)
    src = open(m.__file__).read()
    top = compile(src, m.__file__, "exec")
    for code in walk(top):
        for ins in dis.get_instructions(code):
            if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME") and ins.argval not in m.__dict__ and not hasattr(builtins, ins.argval):
                print(f"{name}: {code.co_name}: undefined global {ins.argval!r} (line {ins.starts_line or code.co_firstlineno})")
                bad += 1
sys.exit(1 if bad else 0)
