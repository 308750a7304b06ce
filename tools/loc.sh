#!/bin/bash
# Lines of product / test / tooling code, INCLUDING *.hip (the prescribed `find` pattern of the verdict omits it).
cd "$(dirname "$0")/.."
count() { find "$@" -type f \( -name '*.hip' -o -name '*.cuh' -o -name '*.hpp' -o -name '*.h' -o -name '*.c' -o -name '*.cpp' -o -name '*.py' -o -name '*.rs' -o -name '*.sh' \) -not -path '*/__pycache__/*' -print0 | xargs -0 cat | wc -l; }
echo "product  (algebra_amd/ include/ rust/ patches/): $(count algebra_amd include rust) + $(cat patches/*.patch | wc -l) patch lines"
echo "oracle/ (test infrastructure):                   $(count oracle)"
echo "tests/:                                          $(count tests)"
echo "tools/ + bench.py + __graft_entry__.py:          $(( $(count tools) + $(cat bench.py __graft_entry__.py | wc -l) ))"
