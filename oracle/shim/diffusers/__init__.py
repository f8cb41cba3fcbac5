__version__ = "0.15.1+shim"
