#!/bin/bash
# usage: tools/lib_ab.sh <other library> [bench flags...]   (GPU box, repository root)
# Two short bench runs with the library in the tree and two with another build of it put in its place, alternating.
set -e
lib=semantic-segmentation_amd/segk/libsegk.so
other=$1; shift
cp $lib /tmp/lib_tree.so
for round in 1 2; do
  echo "== tree"; cp /tmp/lib_tree.so $lib; bash tools/quick_ab.sh "" "$@"
  echo "== other"; cp $other $lib; bash tools/quick_ab.sh "" "$@"
done
cp /tmp/lib_tree.so $lib
