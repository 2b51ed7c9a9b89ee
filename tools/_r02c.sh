python tools/variants.py "cur:" "nobudget:-DDCM_EXP_NOBUDGET" 2>&1 | tail -5
