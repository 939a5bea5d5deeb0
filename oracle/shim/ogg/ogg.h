/* oracle/shim: empty */
