K=tests/test_gpu_keytable.py
D=tests/test_gpu_mldsa.py::test_dilithium_ntt_against_oracle
run() { python -X faulthandler -m pytest "$@" -x -q > /tmp/b.log 2>&1; echo "rc=$? seg=$(grep -c 'Segmentation' /tmp/b.log) $(tail -1 /tmp/b.log | cut -c1-60) :: $*"; }
run $K $D
run $K::test_mlkem_tables_match_the_oracle_call_after_call $K::test_mlkem_table_reports_a_non_canonical_public_key_per_item $K::test_mldsa_table_matches_the_oracle_call_after_call $K::test_mldsa_prepared_private_key_signs_like_the_oracle $D
run $K::test_mldsa_table_of_prepared_private_keys_signs_like_the_oracle $K::test_hybrid_key_tables_match_the_oracle_call_after_call $K::test_batch_public_from_private_keys $D
run $K::test_mlkem_tables_match_the_oracle_call_after_call $K::test_hybrid_key_tables_match_the_oracle_call_after_call $D
run $K::test_mlkem_tables_match_the_oracle_call_after_call $K::test_mldsa_table_of_prepared_private_keys_signs_like_the_oracle $D
run $K::test_mlkem_tables_match_the_oracle_call_after_call $K::test_batch_public_from_private_keys $D
which gdb valgrind 2>&1 | head -2
