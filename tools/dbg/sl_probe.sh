#!/bin/bash
# sidelink self-tests of the reference's sync_sl_test (linked against the product) over PRB counts, with and without standard rates
B=tests/ref_link/_build/bin_full
for args in "-p 6 -c 0 -d" "-p 15 -c 84 -d" "-p 25 -c 168" "-p 25 -c 168 -d" "-p 50 -c 168 -d" "-p 50 -c 168" "-p 75 -c 10 -d" "-p 100 -c 10 -d" "-p 100 -c 10"; do
  echo "== sync_sl_test $args"; $B/sync_sl_test $args 2>&1 | grep -v "^$" | tail -4; echo "rc=$?"
done
