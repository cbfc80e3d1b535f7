#!/bin/bash
# Makes the reference's OWN definitions of the four ORBmatcher member functions that host/ORBmatcherHip.cc replaces
# WEAK inside an already compiled object file, so that src/ORBmatcher.cc needs no edit: link the object beside
# ORBmatcherHip.o and the HIP-backed (strong) definitions win; the other matcher methods keep their CPU bodies.
#   usage: tools/weaken_matcher_symbols.sh path/to/ORBmatcher.cc.o
set -e
obj=${1:?object file}
args=()
while read -r m; do
    case "$(echo "$m" | c++filt)" in
        "ORB_SLAM2::ORBmatcher::DescriptorDistance("*|"ORB_SLAM2::ORBmatcher::SearchForInitialization("*|"ORB_SLAM2::ORBmatcher::SearchByBoW("*)
            args+=("--weaken-symbol=$m");;
    esac
done < <(nm --defined-only "$obj" | awk '$2 == "T" {print $3}')
[ ${#args[@]} -eq 4 ] || { echo "expected 4 symbols to weaken, found ${#args[@]}" >&2; exit 1; }
objcopy "${args[@]}" "$obj"
echo "weakened ${#args[@]} symbols in $obj"
