"""Test/bench tooling (synthetic APT recordings).  Not on the product path."""
