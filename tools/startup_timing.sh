cd $GRAFT_REPO_ROOT
echo "== minimal HIP program (hipMalloc + 1 kernel + sync)"; for i in 1 2 3; do ( time ./tools/ubench/hipmin ) 2>&1 | grep real; done
echo "== bin/m_translate on the tiny golden job"; cd tests/golden/translate_typical; for i in 1 2 3; do ( time ../../../bin/m_translate profiles-l profiles-r nucmer.list /tmp/o.delta ) 2>&1 | grep real; done
echo "== upstream m_translate on the tiny golden job"; for i in 1 2; do ( time ../../../oracle/_ref/m_translate profiles-l profiles-r nucmer.list /tmp/o2.delta ) 2>&1 | grep real; done
