#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# What must hold at every commit (a kernel commit landed after the last CPU run and left HEAD red): the library
# builds from the tracked sources, the archived experiment patches still apply to them, the async-fragment ISA check passes
# (the Makefile runs it on the linked objects' assembly), and the CPU suite is green.  Installed as .git/hooks/pre-commit by
# `tools/precommit.sh --install`; run by hand otherwise.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "--install" ]; then
    printf '#!/bin/bash\nexec "%s/tools/precommit.sh"\n' "$root" > "$root/.git/hooks/pre-commit"
    chmod +x "$root/.git/hooks/pre-commit"
    echo "installed .git/hooks/pre-commit"
    exit 0
fi
cd "$root"
make -s -C poweflownet_amd/csrc -j 8
python -m pytest tests/ -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
test "${PIPESTATUS[0]}" -eq 0
