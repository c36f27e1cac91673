"""Import-path shim for the reference's `utils.*` module names (see fabric_amd/utils)."""
