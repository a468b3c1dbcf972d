from .cli import main

if __name__ == "__main__":  # (row-shard writer processes re-import this module under another name)
    main()
