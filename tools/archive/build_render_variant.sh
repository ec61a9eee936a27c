#!/bin/bash
# kept for the round-4 scripts that call it: tools/build_variant.sh does the work
exec "$(dirname "$0")/build_variant.sh" "$@"
