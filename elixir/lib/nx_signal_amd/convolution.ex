defmodule NxSignalAMD.Convolution do
  @moduledoc """
  The FFT leg of `NxSignal.Convolution` (lib/nx_signal/convolution.ex:38-58, :252-329) for 1-D operands:
  `convolve(a, b, method: :fft)` / `fftconvolve/3`.  Real operands run the overlap-save kernel (any length, batched over
  leading axes of `a`); complex operands one transform of up to 8192 points.  `method: :direct` and n-D operands are not
  part of the accelerated path and raise.
  """
  alias NxSignalAMD.NIF

  @modes %{full: 0, same: 1, valid: 2}

  def convolve(in1, in2, opts \\ []) do
    opts = Keyword.validate!(opts, mode: :full, method: :direct)
    mode!(opts[:mode])

    case opts[:method] do
      :fft -> fftconvolve(in1, in2, mode: opts[:mode])
      :direct -> raise ArgumentError, "method: :direct is not accelerated; use method: :fft or NxSignal.Convolution"
      other -> raise ArgumentError, "expected method to be one of [:direct, :fft], got: #{inspect(other)}"
    end
  end

  def fftconvolve(in1, in2, opts \\ []) do
    opts = Keyword.validate!(opts, [:method, mode: :full])
    mode = mode!(opts[:mode])

    if Nx.rank(in2) != 1 or Nx.rank(in1) < 1 do
      raise ArgumentError, "the accelerated fftconvolve takes a 1-D second operand (the filter); n-D convolution is not accelerated"
    end

    complex? = match?({:c, _}, Nx.type(in1)) or match?({:c, _}, Nx.type(in2))
    ctx = NxSignalAMD.context()

    if complex? do
      if Nx.rank(in1) != 1, do: raise(ArgumentError, "complex fftconvolve takes 1-D operands")
      a = in1 |> Nx.as_type(:c64) |> Nx.to_binary()
      b = in2 |> Nx.as_type(:c64) |> Nx.to_binary()
      {:ok, out} = NIF.fftconvolve_c64(ctx, a, b, mode) |> NxSignalAMD.unwrap!()
      Nx.from_binary(out, :c64)
    else
      {batch_shape, length} = NxSignalAMD.split_last(Nx.shape(in1))
      batch = Tuple.product(batch_shape)
      x = in1 |> Nx.as_type(:f32) |> Nx.to_binary()
      h = in2 |> Nx.as_type(:f32) |> Nx.to_binary()
      {:ok, y} = NIF.fir(ctx, x, length, batch, h, mode) |> NxSignalAMD.unwrap!()
      n_out = div(byte_size(y), 4 * max(batch, 1))
      Nx.from_binary(y, :f32) |> Nx.reshape(Tuple.insert_at(batch_shape, tuple_size(batch_shape), n_out))
    end
  end

  defp mode!(mode) when is_map_key(@modes, mode), do: @modes[mode]

  defp mode!(mode),
    do: raise(ArgumentError, "expected mode to be one of [:full, :same, :valid], got: #{inspect(mode)}")
end
